#!/bin/bash
# Hardware counters of the post-chain kernels (blur X alone, blur Y + tonemap alone at 4K): one rocprofv3 --pmc pass per counter group
# over scripts/bench_post.py with few launches; prints per-kernel medians.   usage (on the GPU box): bash scripts/pmc_post.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export VQ_POST_PARTS=1 VQ_POST_PARTS_ONLY=1 VQ_POST_REPS=20 VQ_POST_SPIN=20
TAG=${1:-post}
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$TAG/$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/$i -- python scripts/bench_post.py > /dev/null 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, statistics as st
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        if "blur" not in name: continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print(f"   {c:40s} {st.median(vals):16.1f}   (n={len(vals)})")
PY
