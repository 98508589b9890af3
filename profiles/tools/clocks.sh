#!/bin/bash
# Samples the GPU's clock and power while the front-end kernel runs back to back: profiles/tools/clocks.sh <lib.so> [flavour]
lib=${1:-sound-event-detection_amd/sed_amd/libsedhip.so}
cfg=${2:-htsat}
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket|Average|Package|power \(" | head -6
timeout -k 10 120 profiles/tools/build/kbench -c $cfg -n 20000 -w 100 $lib > /tmp/kb.out 2>&1 &
pid=$!
sleep 1.0
for i in 1 2 3 4; do
  echo "== under load, sample $i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket|Average|Package|power \(" | head -6
  sleep 0.5
done
wait $pid
cat /tmp/kb.out
