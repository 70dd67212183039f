#!/bin/bash
# Round-3 GPU session 9: forms of the load-time kernels (shared-H BRDF LUT, branch-free diffuse tap): parity, form identity, timings.
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_forms.py -q -x > $O/conv_forms_tests.log 2>&1; echo "conv forms rc=$?"; tail -5 $O/conv_forms_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "lut or conv or cfg4 or ibl or golden" > $O/conv_parity_tests.log 2>&1; echo "conv parity rc=$?"; tail -5 $O/conv_parity_tests.log
timeout 600 python scripts/bench_ibl_forms.py > $O/ibl_forms.jsonl 2> $O/ibl_forms.err; echo "ibl forms rc=$?"; cat $O/ibl_forms.jsonl
