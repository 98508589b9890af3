#!/bin/bash
set -u
tag=${1:-m}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-60)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- $BENCH > "$out/pmc_$name.log" 2>&1 || { echo "pmc $grp failed"; tail -5 "$out/pmc_$name.log"; exit 1; }
done
python3 profiles/summarize.py $out $out/summary > /dev/null 2>&1
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in d["pmc"].items():
    if "stft_mel" in k:
        for c,x in v["mean_per_dispatch"].items(): print(f"{c:34s} {x:.4g}")
PY
