#!/bin/bash
# Round-3 GPU session 16: diffuse convolution on footprint records (default): parity, form identity, timings, counters.
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_forms.py -q > $O/conv_forms_tests.log 2>&1; echo "conv forms rc=$?"; tail -3 $O/conv_forms_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_fixtures.py -q -k "lut or conv or cfg4 or ibl or golden or diffuse" > $O/conv_parity_tests.log 2>&1; echo "conv parity rc=$?"; tail -3 $O/conv_parity_tests.log
timeout 600 python scripts/bench_ibl_forms.py > $O/ibl_forms.jsonl 2> $O/ibl_forms.err; echo "ibl forms rc=$?"; cat $O/ibl_forms.jsonl
bash scripts/pmc_conv.sh records > $O/pmc_conv_records.txt 2>&1; grep -A30 conv_diffuse $O/pmc_conv_records.txt
