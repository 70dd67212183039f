#!/bin/bash
# Round-3 GPU session 6: counters of the final shade kernel, full -m gpu suite, the bench line (default K), the bench under rocprofv3.
O=gpurun_out/r3f; mkdir -p $O
VQ_COMMIT=${VQ_COMMIT:-unknown} bash scripts/pmc_refresh.sh > $O/pmc_refresh.log 2>&1; tail -6 $O/pmc_refresh.log | cut -c1-300
cp gpurun_out/pmc_constants.json profiles/pmc_constants.json; cp gpurun_out/pmc_constants.json $O/pmc_constants.json
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"; cut -c1-300 $O/bench_cfg3.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg3_driver_args.json 2> /dev/null; echo "bench(driver args) rc=$?"
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f/prof -- python bench.py --no-cpu-baseline --no-second-mode > $O/bench_under_profiler.json 2> $O/prof.err; echo "prof rc=$?"
find gpurun_out/r3f/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r3f/prof gpurun_out/pmc
