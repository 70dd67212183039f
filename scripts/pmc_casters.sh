#!/bin/bash
# Counters of the spot-light + PCF caster path (k_forward_lighting<noenv,casters,RGBA16F>) on the two caster workloads of benchlib/casters.py (run on the GPU box):
# VALU instructions per wave, L1 (TCP) lookups, texture-addresser busy, wait cycles, HBM-side traffic — one rocprofv3 pass per counter group.
# usage: bash scripts/pmc_casters.sh <tag>      -> gpurun_out/pmc_casters_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}
OUT=gpurun_out/pmc_casters_$TAG.txt
: > $OUT
mkdir -p gpurun_out/pmc_casters
for cfg in cfg1 engine_max; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
    i=$((i+1)); d=gpurun_out/pmc_casters/${cfg}_$i; rm -rf $d
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python scripts/run_casters_once.py $cfg 4 > /dev/null 2>$d.err
    echo "$cfg group $i rc=$? : $grp"
  done
done
python - <<PY >> $OUT
import csv, glob, statistics as st, collections
for cfg in ("cfg1", "engine_max"):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_casters/{cfg}_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_forward_lighting" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = []
    for f in glob.glob(f"gpurun_out/pmc_casters/{cfg}_1/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_forward_lighting" in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"== {cfg}: kernel duration median {st.median(durs) / 1e3 if durs else float('nan'):.1f} us over {len(durs)} launches (under the profiler, counter pass 1)")
    for k in sorted(vals):
        print(f"{k:32s} median {st.median(vals[k]):18.0f}  (n={len(vals[k])})")
    if vals.get("SQ_INSTS_VALU") and vals.get("SQ_WAVES"):
        print(f"VALU instructions per wave: {st.median(vals['SQ_INSTS_VALU']) / st.median(vals['SQ_WAVES']):.1f}; VMEM reads per wave: {st.median(vals['SQ_INSTS_VMEM_RD']) / st.median(vals['SQ_WAVES']):.1f}")
    if vals.get("FETCH_SIZE") and vals.get("WRITE_SIZE"):
        print(f"HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB): {(2 * st.median(vals['FETCH_SIZE']) + st.median(vals['WRITE_SIZE'])) * 1024 / 1e6:.1f} MB")
PY
cat $OUT
