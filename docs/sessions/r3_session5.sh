#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
VQ_PSMAIN_WAVES_SWEEP=4,5,6 timeout 600 python scripts/bench_psmain.py > $O/psmain.jsonl 2> $O/psmain.err; echo "psmain rc=$?"; cut -c1-420 $O/psmain.jsonl; tail -2 $O/psmain.err
