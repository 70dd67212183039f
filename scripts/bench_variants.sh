#!/bin/bash
# A/B prebuilt library variants (scripts/variants/libvqhip_*.so) through bench.py; usage: bench_variants.sh v1 v2 ...
cp vqengine_amd/lib/libvqhip.so /tmp/base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so vqengine_amd/lib/libvqhip.so; else cp scripts/variants/libvqhip_$v.so vqengine_amd/lib/libvqhip.so; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-second-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['stages'])"
done
cp /tmp/base.so vqengine_amd/lib/libvqhip.so
