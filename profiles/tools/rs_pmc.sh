#!/bin/bash
# Instruction counts and activity of the resampler kernel for one rate pair: profiles/tools/rs_pmc.sh [orig target]
# (run from the repo root on the GPU box; one rocprofv3 pass per counter group, the program itself behind "--").
set -u
out=gpurun_out/rs_pmc
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- python3 $root/profiles/tools/resample_one.py "$@" > "$out/pmc_$name.log" 2>&1 || { echo "pmc $grp failed"; tail -5 "$out/pmc_$name.log"; exit 1; }
done
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $root/profiles/tools/resample_one.py "$@" > "$out/trace.log" 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/pmc_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "resample" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c,v in sorted(acc.items()): print(f"  {c:24s} {sum(v)/len(v):.4g}")
for f in glob.glob("$out/trace/**/*_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "resample" in row["Name"]: print(row["Name"][:60], row["Calls"], row["AverageNs"])
PY
