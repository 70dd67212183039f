#!/bin/bash
# Round-3 GPU session 13: bench line with the `sustained` companion; the bench-flow tests (1 rank, 3 ranks sharing the GPU).
O=gpurun_out/r3m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_flow.py -q -x > $O/bench_flow_tests.log 2>&1; echo "bench flow rc=$?"; tail -3 $O/bench_flow_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3m/bench_driver_args.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "sustained", "frame_latency_ms")})
print(d["cold_start"], d["ibl_load"]["warm_total_ms"], d["ibl_load"]["total_ms"])
PY
