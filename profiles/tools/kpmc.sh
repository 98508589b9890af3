#!/bin/bash
# Instruction counts and activity of the fused kernel in one build: profiles/tools/kpmc.sh <lib.so> [kbench args]
# (run from the repo root on the GPU box; one rocprofv3 pass per counter group).
set -u
lib=$1; shift
tag=$(basename $lib .so)
out=gpurun_out/kpmc_$tag
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout -s KILL 90 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- $root/profiles/tools/build/kbench -n 3 -w 2 "$@" $lib > "$out/pmc_$name.log" 2>&1 || { echo "pmc $grp failed"; tail -5 "$out/pmc_$name.log"; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/pmc_*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "stft_mel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("$tag")
for c,v in sorted(acc.items()): print(f"  {c:24s} {sum(v)/len(v):.4g}")
PY
