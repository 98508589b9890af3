#!/bin/bash
# Time AND shader clock of the ablated builds of profiles/tools/ablate.sh (which phase costs power, not only time):
#   profiles/tools/ablate.sh build ; then on the GPU box: profiles/tools/ablate_clocks.sh
here=$(cd "$(dirname "$0")" && pwd)
for m in ${MASKS:-0 1 2 4 8 16 32 64 96 256 512 768}; do
  $here/build/kbench -n 12000 -w 100 $here/build/libsedhip_ab$m.so > /tmp/abc.out 2>&1 &
  pid=$!
  sleep 1.3
  c1=$(rocm-smi --showclocks 2>/dev/null | grep sclk | head -1 | sed 's/.*(\(.*\)Mhz).*/\1/')
  sleep 0.4
  c2=$(rocm-smi --showclocks 2>/dev/null | grep sclk | head -1 | sed 's/.*(\(.*\)Mhz).*/\1/')
  wait $pid
  t=$(grep -o "best *[0-9.]* us  mean *[0-9.]* us" /tmp/abc.out | head -1)
  echo "mask $m: $t  sclk $c1 $c2 MHz"
done
