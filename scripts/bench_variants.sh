#!/bin/bash
# A/B the shade kernel across prebuilt library variants (scripts/variants/libvqhip_*.so), interleaved, 2 rounds
cp vqengine_amd/lib/libvqhip.so /tmp/base.so
for round in 1; do
  for v in base mad abl1; do
    if [ $v = base ]; then cp /tmp/base.so vqengine_amd/lib/libvqhip.so; else cp scripts/variants/libvqhip_$v.so vqengine_amd/lib/libvqhip.so; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['stages'])"
  done
done
cp /tmp/base.so vqengine_amd/lib/libvqhip.so
