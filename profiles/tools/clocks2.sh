#!/bin/bash
# sclk under a pure v_pk_fma_f32 load of 1, 4 and 12 wavefronts per CU
for w in 1 4 12; do
  profiles/tools/build/ubench10 $w 2.5 > /tmp/u10.out 2>&1 &
  pid=$!
  sleep 1.2
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  sleep 0.5
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  wait $pid
  cat /tmp/u10.out
done
