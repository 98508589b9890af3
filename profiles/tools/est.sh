#!/bin/bash
# Upper-bound estimate for "two groups per wavefront" (8 wavefronts per CU, table reads shared by two groups, the span
# of 8 frames fetched once): timing-only builds (mkvar.sh est*: SED_X_MASK=2 drops the table reads, SED_SPAN_HOPS=4
# fetches 1280 instead of 1984 samples per group) at 12, 10 and 8 wavefronts per CU.  Outputs of these builds are wrong.
b=profiles/tools/build
for w in 12 10 8; do
  echo "== SED_DEBUG_WAVES=$w"
  SED_DEBUG_WAVES=$w timeout -k 10 200 $b/kbench -n 100 $b/libsedhip_est0.so $b/libsedhip_est2.so $b/libsedhip_esth.so $b/libsedhip_est2h.so || exit 1
done
echo "== 256 VGPRs, SED_DEBUG_WAVES=8"
SED_DEBUG_WAVES=8 timeout -k 10 200 $b/kbench -n 100 $b/libsedhip_est0w2.so $b/libsedhip_est2hw2.so || exit 1
echo "== production build"
timeout -k 10 100 $b/kbench -n 100 $b/libsedhip_base.so
