#!/bin/bash
# Hardware counters of the widened per-frame kernels at 4K (G-buffer producer, Z pre-pass normals, SSR fallback): one rocprofv3 --pmc pass per counter group over scripts/run_widened_once.py;
# prints per-kernel medians.   usage (on the GPU box): bash scripts/pmc_widened.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-widened}
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TA_BUSY_avr" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD" \
           "TA_ADDR_STALL_BY_TC_CYCLES_sum TA_DATA_STALL_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$TAG/$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/$i -- python scripts/run_widened_once.py > /dev/null 2> gpurun_out/pmc_$TAG.err$i
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, statistics as st
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        if not any(k in name for k in ("gbuffer_from_materials", "scene_normals", "ssr_env")): continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print(f"   {c:40s} {st.median(vals):16.1f}   (n={len(vals)})")
PY
rm -rf gpurun_out/pmc_$TAG
