#!/bin/bash
# VALU instructions per wave of the shade kernel of the CURRENT build (run on the GPU box): rocprofv3 PMC pass over bench.py.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/pmc_shade
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_shade -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
python - <<PY
import csv, glob, statistics as st
f = glob.glob("gpurun_out/pmc_shade/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_forward_lighting" in r["Kernel_Name"]]
valu = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_INSTS_VALU"]
waves = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_WAVES"]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Counter_Name"] == "SQ_WAVES"]
print("k_forward_lighting: launches", len(valu), "valu/wave", round(st.median(valu) / st.median(waves), 1), "median dur_us", st.median(dur) / 1e3)
PY
