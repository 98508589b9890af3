#!/bin/bash
# What the Kaldi flavour pays for its frame conditioning and its 128 mel bins (kbench, same call):
# full flavour, without DC removal, without pre-emphasis, without both (the plain kernel with Kaldi's window and mel), 64 mel bins.
b=profiles/tools/build
lib=sound-event-detection_amd/sed_amd/libsedhip.so
for args in "" "-P 2" "-P 1" "-P 0" "-M 64" "-P 0 -M 64"; do
  echo "== kaldi $args"
  timeout -k 10 120 $b/kbench -c kaldi -n 100 $args $lib || exit 1
done
echo "== yamnet"; timeout -k 10 120 $b/kbench -c yamnet -n 100 $lib
