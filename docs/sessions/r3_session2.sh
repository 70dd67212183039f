#!/bin/bash
# Round-3 GPU session 2: the new bench line (N = 1, all extras), bench-flow tests, in-process shade A/B, full -m gpu suite.
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-400 $O/bench_n1.json; tail -3 $O/bench_n1.err
timeout 300 python scripts/ab_shade.py scripts/variants/libvqhip_r2.so noise > $O/ab_shade.jsonl 2> $O/ab_shade.err; timeout 300 python scripts/ab_shade.py scripts/variants/libvqhip_r2.so coherent >> $O/ab_shade.jsonl 2>> $O/ab_shade.err; cat $O/ab_shade.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -8 $O/gpu_tests.log
