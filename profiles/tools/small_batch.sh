#!/bin/bash
# Few streams per launch (serving at low load): profiles/tools/small_batch.sh libA.so [libB.so]
for c in htsat yamnet crnn; do
  for b in 1 2 4 8 16 32 64; do
    timeout -k 10 120 profiles/tools/build/kbench -c $c -b $b -n 200 "$@" || exit 1
  done
done
