#!/bin/bash
# Round-3 GPU session 11: short form of pow in the direct tonemapper (HDR default, RGBA32F images): parity + A/B against the previous build.
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -k "tonemap or post or blur or golden" > $O/post_tests.log 2>&1; echo "post tests rc=$?"; tail -4 $O/post_tests.log
timeout 300 python scripts/bench_hdr_post.py > $O/hdr_post.jsonl 2> $O/hdr_post.err; echo "rc=$?"
VQHIP_LIBRARY_PATH=$PWD/scripts/variants/prev/libvqhip_prev.so timeout 300 python scripts/bench_hdr_post.py >> $O/hdr_post.jsonl 2>> $O/hdr_post.err; echo "rc=$?"
cat $O/hdr_post.jsonl; tail -3 $O/hdr_post.err
