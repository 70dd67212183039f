#!/bin/bash
# Round-3 GPU session 17: shade workgroup size by frame size (128 lanes under 4 Mpixel): parity, A/B, counter refresh, bench line.
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python scripts/bench_shade_wg.py > $O/shade_wg.jsonl 2> /dev/null; cat $O/shade_wg.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -a 'passed\|failed' $O/gpu_tests.log | tail -1
VQ_COMMIT=${VQ_COMMIT:-unknown} bash scripts/pmc_refresh.sh > $O/pmc_refresh.log 2>&1; tail -4 $O/pmc_refresh.log | cut -c1-300
cp gpurun_out/pmc_constants.json $O/pmc_constants.json
