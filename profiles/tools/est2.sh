#!/bin/bash
# Timing-only estimates: (1) HTS-AT with 40 instead of 48 matrix instructions in the mel stage (unequal chain lengths
# per accumulator set would allow U = 6 + 4 steps: SED_EST_MEL_U=5 stands in for the average); (2) the n_fft 2048 flavours
# at 4 and 5 wavefronts per CU (what a 6th wavefront is worth).
b=profiles/tools/build
timeout -k 10 200 $b/kbench -n 100 $b/libsedhip_est0.so $b/libsedhip_estu5.so $b/libsedhip_est0.so $b/libsedhip_estu5.so || exit 1
for c in crnn dbmel; do
  for w in 3 4 5; do
    echo "== $c SED_DEBUG_WAVES=$w"
    SED_DEBUG_WAVES=$w timeout -k 10 200 $b/kbench -c $c -n 100 $b/libsedhip_fulldiag.so || exit 1
  done
done
