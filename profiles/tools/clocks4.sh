#!/bin/bash
# Is the lower clock under the fused kernel a matter of short launches?  Pure packed-FMA load, 12 wavefronts per CU,
# as 28 ms launches and as back-to-back launches of about 150 us (the fused kernel's length).
for it in 20000 110; do
  profiles/tools/build/ubench10 12 2.5 $it > /tmp/u10.out 2>&1 &
  pid=$!
  sleep 1.2
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  sleep 0.5
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  wait $pid
  cat /tmp/u10.out
done
