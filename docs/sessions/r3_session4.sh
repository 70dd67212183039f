#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_gbuffer.py tests/test_gpu_parity.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 600 python scripts/bench_psmain.py > $O/psmain.jsonl 2> $O/psmain.err; echo "psmain rc=$?"; cat $O/psmain.jsonl; tail -2 $O/psmain.err
timeout 300 python scripts/bench_gbuffer.py 2>/dev/null | head -3 > $O/gbuffer.jsonl; cut -c1-300 $O/gbuffer.jsonl
timeout 300 python scripts/ab_shade.py scripts/variants/libvqhip_oldshade.so noise > $O/ab_shade.jsonl 2> $O/ab_shade.err; cat $O/ab_shade.jsonl
