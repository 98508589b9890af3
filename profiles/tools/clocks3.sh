#!/bin/bash
# sclk under the synthetic loops of ubench8: arithmetic alone, LDS traffic alone, both
for m in v l x1; do
  profiles/tools/build/ubench8 $m 2.5 > /tmp/u8.out 2>&1 &
  pid=$!
  sleep 1.2
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  sleep 0.5
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1
  wait $pid
  cat /tmp/u8.out
done
