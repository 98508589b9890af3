#!/bin/bash
# Round-3 evidence in one call on the GPU box (from the repo root): rocprofv3 trace + PMC passes of bench.py, the bench
# lines of the three workloads, the flavour table, the side-stream A/B with the collective forced on a one-rank RCCL
# group, and the clock under load.  Everything lands in gpurun_out/.
set -u
o=gpurun_out
profiles/collect.sh r03 > $o/collect_r03.log 2>&1 || { tail -5 $o/collect_r03.log; exit 1; }
python3 profiles/summarize.py $o/prof_r03 $o/r03_summary > /dev/null 2>&1 || exit 1
timeout -k 10 300 python3 bench.py > $o/r03_bench_htsat.json 2> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/r03_bench_driverstyle.json 2>> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --workload yamnet > $o/r03_bench_yamnet.json 2>> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --workload crnn_chain > $o/r03_bench_crnn_chain.json 2>> $o/r03_bench_err.log || exit 1
profiles/tools/flavours.sh > $o/r03_flavours.log 2>&1 || exit 1
for rep in 1 2; do
  for mode in same side; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --decode-stream $mode > $o/r03_ab_${mode}_plain_$rep.json 2>> $o/r03_bench_err.log || exit 1
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --decode-stream $mode --force-collective > $o/r03_ab_${mode}_rccl_$rep.json 2>> $o/r03_bench_err.log || exit 1
  done
done
profiles/tools/clocks.sh > $o/r03_clocks.log 2>&1
echo done
