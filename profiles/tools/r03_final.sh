#!/bin/bash
# Final numbers of round 3 in one call: round-2 library against the round-3 library on every flavour (one process per
# flavour, outputs compared), the GPU tests, the smoke check, the rocprofv3 trace + PMC passes of the bench command, the bench
# lines of every workload, the hand-off timings and the clock under load.  Everything lands in gpurun_out/.
set -u
o=gpurun_out
profiles/tools/ab_all.sh profiles/tools/build/libsedhip_r02.so sound-event-detection_amd/sed_amd/libsedhip.so profiles/tools/build/libsedhip_r02.so sound-event-detection_amd/sed_amd/libsedhip.so > $o/r03_ab_r02.log 2>&1 || exit 1
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $o/r03_gputest.log 2>&1 || { tail -20 $o/r03_gputest.log; exit 1; }
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $o/r03_smoke.log 2>&1 || exit 1
profiles/collect.sh r03 > $o/collect_r03.log 2>&1 || { tail -5 $o/collect_r03.log; exit 1; }
python3 profiles/summarize.py $o/prof_r03 $o/r03_summary > /dev/null 2>&1 || exit 1
timeout -k 10 300 python3 bench.py > $o/r03_bench_htsat.json 2> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $o/r03_bench_driverstyle.json 2>> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --workload yamnet > $o/r03_bench_yamnet.json 2>> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --workload crnn_chain > $o/r03_bench_crnn_chain.json 2>> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --workload server --steps 200 --warmup 20 > $o/r03_bench_server.json 2>> $o/r03_bench_err.log || exit 1
timeout -k 10 300 python3 bench.py --gpus 2 --one-device --backend gloo --steps 200 --warmup 20 --no-cpu-baseline > $o/r03_bench_2ranks_gloo.json 2>> $o/r03_bench_err.log || exit 1
profiles/tools/flavours.sh > $o/r03_flavours.log 2>&1 || exit 1
timeout -k 10 300 python3 profiles/tools/seg_bench.py > $o/r03_seg_bench.log 2>&1 || exit 1
profiles/tools/clocks.sh > $o/r03_clocks.log 2>&1
echo done
