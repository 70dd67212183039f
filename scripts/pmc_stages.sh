#!/bin/bash
# VALU instructions per wave of every kernel exercised by scripts/bench_stages.py (run on the GPU box).
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf gpurun_out/pmc_stages
timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_stages -- python scripts/bench_stages.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections, statistics as st
f = glob.glob("gpurun_out/pmc_stages/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:52]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "SQ_WAVES":
        acc[name]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in acc.items():
    if k.startswith("__amd") or k.startswith("at::"): continue
    groups = collections.defaultdict(list)
    for i, w in enumerate(v["SQ_WAVES"]): groups[w].append(i)
    for w, idx in sorted(groups.items()):
        if len(idx) < 3: continue
        valu = st.median([v["SQ_INSTS_VALU"][i] for i in idx]); d = st.median([v["dur"][i] for i in idx])
        print(f"{k:54s} waves={w:9.0f} n={len(idx):3d} valu/wave={valu / w:10.1f} dur_us={d / 1e3:9.1f} lane-instr/s={valu * 64 / (d * 1e-9) / 1e12:5.1f}T")
PY
