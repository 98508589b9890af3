#!/bin/bash
# quick PMC pass: instruction counts and activity of the fused kernel
set -u
tag=${1:-q}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-60)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- $BENCH > "$out/pmc_$name.log" 2>&1 || { echo "pmc $grp failed"; tail -5 "$out/pmc_$name.log"; exit 1; }
done
python3 profiles/summarize.py $out $out/summary > /dev/null 2>&1
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d["pmc"].items():
    if "stft_mel" in k:
        for c,x in v["mean_per_dispatch"].items(): print(f"{c:28s} {x:.4g}")
PY
