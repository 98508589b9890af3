#!/bin/bash
# HBM traffic of one launch of every front-end flavour: two rocprofv3 passes each (FETCH_SIZE, WRITE_SIZE, never combined
# with a trace), kbench as the program.  Run from the repo root on the GPU box: profiles/tools/traffic_all.sh [lib.so]
set -u
lib=${1:-sound-event-detection_amd/sed_amd/libsedhip.so}
out=gpurun_out/traffic
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in htsat htsat_i16 yamnet kaldi crnn dbmel htsat1h; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d "$out/${c}_$ctr" -- $root/profiles/tools/build/kbench -c $c -n 5 -w 2 $root/$lib > "$out/${c}_$ctr.log" 2>&1 || { echo "$c $ctr failed"; tail -5 "$out/${c}_$ctr.log"; exit 1; }
  done
done
python3 - <<PY
import csv, glob, json
res = {}
for c in "htsat htsat_i16 yamnet kaldi crnn dbmel htsat1h".split():
    ent = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("$out/%s_%s/**/*_counter_collection.csv" % (c, ctr), recursive=True):
            for row in csv.DictReader(open(f)):
                if "stft_mel" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    vals.append(float(row["Counter_Value"]))
        ent[ctr] = sum(vals) / max(len(vals), 1)
        ent["n_" + ctr] = len(vals)
    res[c] = ent
json.dump(res, open("$out/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
