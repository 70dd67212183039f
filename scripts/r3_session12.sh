#!/bin/bash
# Round-3 GPU session 12: software-pipelined diffuse tap loop (A/B), form identity.
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_forms.py -q -x > $O/conv_forms_tests.log 2>&1; echo "conv forms rc=$?"; tail -3 $O/conv_forms_tests.log
VQHIP_DIFFUSE_FORM=pipelined timeout 900 python -m pytest tests/test_gpu_conv_forms.py tests/test_gpu_parity.py -q -k "diffuse or conv or cfg4" > $O/conv_pipelined_tests.log 2>&1; echo "pipelined parity rc=$?"; tail -3 $O/conv_pipelined_tests.log
timeout 600 python scripts/bench_ibl_forms.py > $O/ibl_forms.jsonl 2> $O/ibl_forms.err; echo "ibl forms rc=$?"; grep diffuse $O/ibl_forms.jsonl
