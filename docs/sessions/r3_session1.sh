#!/bin/bash
# Round-3 GPU session 1: parity of the new kernel forms, forms of the Y pass, shade on the three frames (new library vs the round-2 one).
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q > $O/tests_round3.log 2>&1; echo "round3 tests rc=$?"; tail -3 $O/tests_round3.log
VQ_YFORMS_FRAME="lut64,c8s,c16s,c8sw5,c16" timeout 900 python scripts/bench_yforms.py > $O/yforms.jsonl 2> $O/yforms.err; echo "yforms rc=$?"; cat $O/yforms.jsonl
timeout 600 python scripts/bench_shade_content.py new > $O/shade_content.jsonl 2> $O/shade_content.err; echo "shade new rc=$?"
cp vqengine_amd/lib/libvqhip.so /tmp/new.so; cp scripts/variants/libvqhip_r2.so vqengine_amd/lib/libvqhip.so
timeout 600 python scripts/bench_shade_content.py r2 >> $O/shade_content.jsonl 2>> $O/shade_content.err; echo "shade r2 rc=$?"
cp /tmp/new.so vqengine_amd/lib/libvqhip.so
cat $O/shade_content.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.log
