cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_fixtures.py tests/test_gpu_round3.py tests/test_gpu_arith_modes.py -q -m gpu -x -k "forward or caster or cfg1 or spot or shade or psmain or band" 2>&1 | tail -15
python - <<'PY' > gpurun_out/r6b_casters_after.json 2> gpurun_out/r6b_casters_after.err
import json, sys
sys.path.insert(0, ".")
import bench
from benchlib import casters
from vqengine_amd import capi
ctx = capi.Context(0)
r = casters.casters_report(ctx, bench._stage_stats, bench.HBM_PEAK_GBPS)
print(json.dumps(r))
PY
cat gpurun_out/r6b_casters_after.json; tail -5 gpurun_out/r6b_casters_after.err
