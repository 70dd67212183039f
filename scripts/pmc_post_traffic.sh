#!/bin/bash
# HBM-side traffic of the post kernels at 4K: FETCH_SIZE and WRITE_SIZE, each in its OWN rocprofv3 pass (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64-byte units of half the channels: x2).
# usage (on the GPU box): bash scripts/pmc_post_traffic.sh
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export VQ_SPIN=10 VQ_REPS=4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_pt/$c
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_pt/$c -- python scripts/run_post_once.py > /dev/null 2> gpurun_out/pmc_pt_$c.err
  echo "pass $c rc=$?"
done
python - <<PY
import csv, glob, collections, statistics as st
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_pt/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        if any(k in name for k in ("k_blur_x4", "k_blur_y_tonemap_lut", "k_post_chain")):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
px = 3840 * 2160
for k, v in sorted(acc.items()):
    f, w = st.median(v["FETCH_SIZE"][-4:]), st.median(v["WRITE_SIZE"][-4:])      # KiB
    print(f"{k:34s} FETCH_SIZE {f:12.0f} KiB  WRITE_SIZE {w:12.0f} KiB  ->  HBM-side bytes/launch = 2*FETCH + WRITE = {(2*f+w)*1024/1e6:8.1f} MB = {(2*f+w)*1024/px:6.2f} B/px")
PY
rm -rf gpurun_out/pmc_pt
