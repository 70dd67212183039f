#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench_driver_args.json
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3h/prof -- python bench.py --no-cpu-baseline --no-second-mode --no-extras > $O/bench_under_profiler.json 2> $O/prof.err; echo "prof rc=$?"
find gpurun_out/r3h/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/r3h/prof
python __graft_entry__.py smoke 2>&1 | tail -2
