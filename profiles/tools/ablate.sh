#!/bin/bash
# Ablation of the fused kernel's streaming loop: one diagnostic build per switched-off phase (compile-time mask
# SED_X_MASK, bits in csrc/fft_core.h), timed with kbench on the headline workload.  Outputs of ablated builds are
# wrong by construction; only the time matters.
#   profiles/tools/ablate.sh build     (in the build container: writes profiles/tools/build/libsedhip_ab*.so)
#   profiles/tools/ablate.sh run       (on the GPU box, from the repo root)
here=$(cd "$(dirname "$0")" && pwd)
MASKS="0 1 2 4 8 16 32 64 96 256 512 768 1024"
case "$1" in
  build)
    for m in $MASKS; do $here/mkvar.sh ab$m -DSED_X_MASK=$m > /dev/null || exit 1; done
    $here/mkvar.sh abtl -DSED_TIMELINE > /dev/null && $here/mkvar.sh abst -DSED_STAMPS > /dev/null ;;
  run)
    libs=""
    for m in $MASKS; do libs="$libs $here/build/libsedhip_ab$m.so"; done
    timeout -k 10 300 $here/build/kbench -n 100 $libs || exit 1
    timeout -k 10 60 $here/build/kbench -n 100 $here/build/libsedhip_abtl.so || exit 1
    timeout -k 10 60 $here/build/kbench -n 100 $here/build/libsedhip_abst.so ;;
  *) echo "usage: ablate.sh build|run"; exit 2 ;;
esac
