#!/bin/bash
# Re-measures the counter-derived constants bench.py reports for the shade kernel of the CURRENT build (run on the GPU box):
#   VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES), transcendental (quarter-rate) instructions per wave (SQ_INSTS_VALU_TRANS where the
#   counter exists), HBM bytes per launch (FETCH_SIZE and WRITE_SIZE, each in its OWN rocprofv3 pass; FETCH doubled as
#   /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950) — for cfg3 in both Fresnel-pow modes and for cfg5.
# Writes gpurun_out/pmc_constants.json stamped with the sha256 of the kernel sources; copy it to profiles/pmc_constants.json and commit it.
# bench.py refuses (nulls) the constants when the sources have changed since.   usage: VQ_COMMIT=<sha> bash scripts/pmc_refresh.sh
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export VQ_BENCH_SPINUP=20
run() {   # tag, counters, bench args
  rm -rf gpurun_out/pmc/$1
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc/$1 -- python bench.py --no-cpu-baseline --no-second-mode --no-extras --steps 6 --warmup 2 $3 > /dev/null 2>&1
  echo "pass $1 rc=$?"
}
run cfg3_product_valu  "SQ_INSTS_VALU SQ_WAVES" "--config cfg3"
run cfg3_product_trans "SQ_INSTS_VALU_TRANS"    "--config cfg3"
run cfg3_product_fetch "FETCH_SIZE"             "--config cfg3"
run cfg3_product_write "WRITE_SIZE"             "--config cfg3"
run cfg3_exp2_valu     "SQ_INSTS_VALU SQ_WAVES" "--config cfg3 --fresnel-pow exp2_log2"
run cfg3_exp2_trans    "SQ_INSTS_VALU_TRANS"    "--config cfg3 --fresnel-pow exp2_log2"
run cfg3c_product_valu  "SQ_INSTS_VALU SQ_WAVES" "--config cfg3 --content coherent"
run cfg3c_product_fetch "FETCH_SIZE"             "--config cfg3 --content coherent"
run cfg3c_product_write "WRITE_SIZE"             "--config cfg3 --content coherent"
run cfg5_product_valu  "SQ_INSTS_VALU SQ_WAVES" "--config cfg5"
run cfg5_product_trans "SQ_INSTS_VALU_TRANS"    "--config cfg5"
run cfg5_product_fetch "FETCH_SIZE"             "--config cfg5"
run cfg5_product_write "WRITE_SIZE"             "--config cfg5"
run2() {  # cfg2: the shade kernel alone (scripts/run_cfg2_once.py)
  rm -rf gpurun_out/pmc/$1
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc/$1 -- python scripts/run_cfg2_once.py > /dev/null 2>&1
  echo "pass $1 rc=$?"
}
run2 cfg2_product_valu  "SQ_INSTS_VALU SQ_WAVES"
run2 cfg2_product_fetch "FETCH_SIZE"
run2 cfg2_product_write "WRITE_SIZE"
python - <<PY
import csv, glob, json, os, statistics as st, sys
sys.path.insert(0, os.getcwd())
import bench
def med(tag, counter):
    fs = glob.glob(f"gpurun_out/pmc/{tag}/*/*counter_collection.csv")
    if not fs:
        return None
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == counter and "k_forward_lighting" in r["Kernel_Name"]]
    return st.median(v) if v else None
out = {"kernel_sources_sha256": bench.kernel_source_hash(), "measured_at_commit": os.environ.get("VQ_COMMIT", "unknown"),
       "profile": "scripts/pmc_refresh.sh: rocprofv3 --pmc <counter> --kernel-trace, one pass per counter group, medians over the shade launches of bench.py --steps 6",
       "sources": bench.PMC_SOURCES}
for key, tag in (("cfg3/product", "cfg3_product"), ("cfg3/exp2_log2", "cfg3_exp2"), ("cfg3_coherent/product", "cfg3c_product"), ("cfg5/product", "cfg5_product"), ("cfg2/product", "cfg2_product")):
    valu, waves = med(tag + "_valu", "SQ_INSTS_VALU"), med(tag + "_valu", "SQ_WAVES")
    if not valu or not waves:
        print("no VALU counters for", key); continue
    e = {"valu_instr_per_wave": round(valu / waves, 1), "waves": waves}
    tr = med(tag + "_trans", "SQ_INSTS_VALU_TRANS")
    if tr:
        e["quarter_rate_instr_per_wave"] = round(tr / waves, 1)
    else:   # the counter is not exposed on this pool: ISA count instead — 5 v_rsq/v_rcp per executed light (2 sqrt seeds, 3 reciprocals), ~12 per pixel
        lights = bench.CONFIGS[key.split("/")[0].split("_")[0]]["lights"]
        e["quarter_rate_instr_per_wave"] = round(3 * lights * 0.82 + 8)
        e["quarter_rate_source"] = "estimate: ISA count (3 per executed light: two v_rsq_f32 + one v_rcp_f32; + 8 per pixel) x 0.82 wave-level executed-light fraction; SQ_INSTS_VALU_TRANS is not available"
    f, w = med(tag + "_fetch", "FETCH_SIZE"), med(tag + "_write", "WRITE_SIZE")
    if f and w:
        e["fetch_size_kib"], e["write_size_kib"] = f, w
        e["hbm_bytes_per_launch"] = int((2 * f + w) * 1024)
    out[key] = e
    print(key, e)
# cfg3/exp2_log2 shares the memory behaviour of cfg3/product (same loads, other arithmetic)
if "cfg3/exp2_log2" in out and "hbm_bytes_per_launch" in out.get("cfg3/product", {}):
    out["cfg3/exp2_log2"].setdefault("hbm_bytes_per_launch", out["cfg3/product"]["hbm_bytes_per_launch"])
try:    # the counters of the load-time kernels (scripts/pmc_conv.sh, stamped with their own source hash) ride along unchanged
    out["load_time_kernels"] = json.load(open("profiles/pmc_constants.json"))["load_time_kernels"]
except (OSError, KeyError, ValueError):
    pass
json.dump(out, open("gpurun_out/pmc_constants.json", "w"), indent=1)
print(json.dumps(out))
PY
