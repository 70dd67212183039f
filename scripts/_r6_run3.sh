cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc_casters.sh after 2>&1 | tail -50
python - <<'PY' > gpurun_out/r6g_casters_after.json 2> gpurun_out/r6g_casters_after.err
import json, sys
sys.path.insert(0, ".")
import bench
from benchlib import casters
from vqengine_amd import capi
ctx = capi.Context(0)
r = casters.casters_report(ctx, bench._stage_stats, bench.HBM_PEAK_GBPS, cores=bench.host_cores())
print(json.dumps(r))
PY
cat gpurun_out/r6g_casters_after.json
