#!/bin/bash
# Where the shade kernel's cycles go (run on the GPU box): SQ busy / wave / instruction-issue counters, one rocprofv3 pass per group.
# usage: bash scripts/pmc_shade_util.sh [cfg2]   (cfg2: the 1080p / 16-light launch of scripts/run_cfg2_once.py instead of the cfg3 headline)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
i=0
CMD="python bench.py --no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3"
[ "$1" = "cfg2" ] && CMD="env VQ_CFG2_REPS=200 python scripts/run_cfg2_once.py"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_util_$i
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_util_$i -- $CMD > /dev/null 2>gpurun_out/pmc_util_$i.err
  echo "group $i rc=$? : $grp"
done
python - <<PY
import csv, glob, statistics as st, collections
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_util_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_forward_lighting" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    print(f"{k:24s} median {st.median(vals[k]):16.0f}  (n={len(vals[k])})")
durs = []
for f in glob.glob("gpurun_out/pmc_util_1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_forward_lighting" in r["Kernel_Name"]:
            durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
if durs and vals.get("SQ_BUSY_CYCLES"):
    d = st.median(durs)
    print(f"kernel duration median {d / 1e3:.1f} us; SQ_BUSY_CYCLES / 32 SQ instances / duration = {st.median(vals['SQ_BUSY_CYCLES']) / 32 / d:.3f} GHz")
PY
