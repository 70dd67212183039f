cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in cube1 cube2; do
  cp scripts/variants/libvqhip_$v.so vqengine_amd/lib/libvqhip.so
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_$v -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
  python - <<PY
import csv, glob, statistics as st
f = glob.glob("gpurun_out/pmc_$v/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_forward_lighting" in r["Kernel_Name"]]
valu = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_INSTS_VALU"]
waves = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_WAVES"]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Counter_Name"] == "SQ_WAVES"]
print("$v", "valu/wave", st.median(valu) / st.median(waves), "dur_us", st.median(dur) / 1e3)
PY
done
