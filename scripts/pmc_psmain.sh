#!/bin/bash
# Hardware counters of PSMain's three forms at 4K (producer alone, lighting alone, one kernel): one rocprofv3 --pmc pass per counter group over
# scripts/run_psmain_once.py; prints per-kernel medians.   usage (on the GPU box): [VQ_PSMAIN_WAVES=4] [VQ_OPTIONS=k=v,...] bash scripts/pmc_psmain.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-psmain}
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$TAG/$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/$i -- python scripts/run_psmain_once.py > /dev/null 2> gpurun_out/pmc_$TAG.err$i
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, statistics as st
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        if not any(k in name for k in ("gbuffer_from_materials", "forward_from_materials", "k_forward_lighting")): continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print(f"   {c:40s} {st.median(vals):16.1f}   (n={len(vals)})")
PY
rm -rf gpurun_out/pmc_$TAG
