#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh <tag>
# 1. kernel trace + stats of the bench command; 2. one PMC pass per counter group (never combined with traces).
# Raw output goes to gpurun_out/prof_<tag>/; the summaries to keep are copied by profiles/summarize.py.
set -u
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
# the trace pass runs the default step count, so that its average kernel time is comparable with bench.py's own
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-60)
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$name" -- $BENCH > "$out/pmc_$name.log" 2>&1 || { echo "pmc $grp failed"; tail -5 "$out/pmc_$name.log"; exit 1; }
done
echo "collected into $out"
find "$out" -name "*.csv" | head -40
