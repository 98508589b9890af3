#!/bin/bash
# rocprofv3 kernel trace of every front-end flavour (kbench, 3 x 100 launches after 30 warm-up launches):
#   profiles/tools/flavours_rocprof.sh [lib.so]   -> gpurun_out/prof_flavours/<flavour>/..._kernel_stats.csv
lib=${1:-sound-event-detection_amd/sed_amd/libsedhip.so}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out/prof_flavours
mkdir -p $out
for c in htsat htsat_i16 yamnet kaldi crnn dbmel htsat1h; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$c -- profiles/tools/build/kbench -c $c -n 100 $lib > $out/$c.log 2>&1 || { echo "$c failed"; tail -3 $out/$c.log; exit 1; }
  f=$(find $out/$c -name "*_kernel_stats.csv" | head -1)
  python3 - "$c" "$f" <<'PY'
import csv, sys
c, f = sys.argv[1], sys.argv[2]
for row in csv.DictReader(open(f)):
    if "stft_mel_kernel" in row["Name"]:
        print(f"{c}: calls {row['Calls']} avg {float(row['AverageNs'])/1e3:.1f} us min {float(row['MinNs'])/1e3:.1f} max {float(row['MaxNs'])/1e3:.1f}")
PY
done
