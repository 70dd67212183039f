#!/bin/bash
# HBM traffic of the bench kernels of the CURRENT build from the PMC counters (run on the GPU box): one rocprofv3 pass per
# counter (FETCH_SIZE and WRITE_SIZE in one pass crashed the tool on this pool), each under its own timeout. Unit: KiB.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
  echo "pass $c rc=$?"
done
python - <<PY
import csv, glob, statistics as st, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")
    if not fs:
        print("no output for", c); continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            by[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
    for k, v in by.items():
        res[k][c] = (st.median(v), len(v))
for k, d in sorted(res.items()):
    if any(s in k for s in ("k_forward", "k_blur", "k_tonemap")):
        f, w = d.get("FETCH_SIZE", (0, 0)), d.get("WRITE_SIZE", (0, 0))
        print(f"{k:72s} launches {f[1]:4d} FETCH {f[0]:10.0f} KiB  WRITE {w[0]:10.0f} KiB  traffic (2F+W) {(2*f[0]+w[0])*1024/1e6:9.1f} MB")
PY
