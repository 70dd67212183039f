#!/bin/bash
# Hardware counters of the post kernels at 4K (X pass, fused Y + tonemap, one-kernel chain): one rocprofv3 --pmc pass per counter group over scripts/run_post_once.py;
# prints per-kernel medians of the LAST launches (the first 200 X passes are the spin-up).   usage (on the GPU box): bash scripts/pmc_post.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-post}
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE WRITE_SIZE TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$TAG/$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/$i -- python scripts/run_post_once.py > /dev/null 2> gpurun_out/pmc_$TAG.err$i
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections, statistics as st
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/*/*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        if not any(k in name for k in ("k_blur_x4", "k_blur_y_tonemap_lut", "k_post_chain")): continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    print(k)
    for c, vals in sorted(v.items()):
        tail = vals[-12:]                                    # the launches after the spin-up
        print(f"   {c:40s} {st.median(tail):16.1f}   (n={len(vals)})")
PY
rm -rf gpurun_out/pmc_$TAG
