#!/bin/bash
# A/B prebuilt library variants (scripts/variants/libvqhip_*.so) through scripts/bench_post.py; usage: post_variants.sh v1 v2 ...
cp vqengine_amd/lib/libvqhip.so /tmp/base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so vqengine_amd/lib/libvqhip.so; else cp scripts/variants/libvqhip_$v.so vqengine_amd/lib/libvqhip.so; fi
  echo "== $v"; python scripts/bench_post.py 2>/dev/null | grep -v '"one"' | cut -c1-120
done
cp /tmp/base.so vqengine_amd/lib/libvqhip.so
