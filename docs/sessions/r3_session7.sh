#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "post_process_one_call" > $O/tests_post.log 2>&1; echo "post tests rc=$?"; tail -4 $O/tests_post.log
timeout 300 python -m pytest tests/test_gpu_round3.py -q -k "second_thread or from_materials" > $O/tests_misc.log 2>&1; echo "misc rc=$?"; tail -3 $O/tests_misc.log
timeout 600 python scripts/bench_post.py > $O/post.jsonl 2> $O/post.err; cat $O/post.jsonl | cut -c1-260
for seg in 9 13 17 26 34; do echo "segments $seg"; VQHIP_POST_SEGMENTS=$seg VQ_POST_REPS=100 VQ_POST_SPIN=100 timeout 300 python scripts/bench_post.py 2>/dev/null | grep "one-2" | cut -c1-120; done | tee $O/post_segments.txt
