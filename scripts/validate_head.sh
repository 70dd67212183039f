#!/bin/bash
# Validation of a HEAD on the GPU box: the whole -m gpu suite, the bench line with the driver's arguments, the headline under rocprofv3 --stats, smoke().  usage: VQ_TAG=r3l bash scripts/validate_head.sh
O=gpurun_out/${VQ_TAG:-r3l}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench_driver_args.json
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${VQ_TAG:-r3l}/prof -- python bench.py --no-cpu-baseline --no-second-mode --no-extras > $O/bench_under_profiler.json 2> $O/prof.err; echo "prof rc=$?"
find gpurun_out/${VQ_TAG:-r3l}/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/${VQ_TAG:-r3l}/prof
python __graft_entry__.py smoke 2>&1 | tail -2
