#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
timeout 300 ./scripts/ubench/stream_patterns > $O/stream_patterns.jsonl 2> $O/stream_patterns.err; echo "stream rc=$?"; cat $O/stream_patterns.jsonl
timeout 300 python scripts/ab_shade.py scripts/variants/libvqhip_oldshade.so noise > $O/ab_shade.jsonl 2> $O/ab_shade.err; timeout 300 python scripts/ab_shade.py scripts/variants/libvqhip_oldshade.so coherent >> $O/ab_shade.jsonl 2>> $O/ab_shade.err; cat $O/ab_shade.jsonl; tail -2 $O/ab_shade.err
