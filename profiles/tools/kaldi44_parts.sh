#!/bin/bash
# The Kaldi filterbank at 44.1 kHz (25 ms = 1102 samples in 2048-point transforms every 441 samples, 128 mel bins): what the
# hop off the 16-byte grid, the frame conditioning and the mel count cost (kbench, same call).
b=profiles/tools/build
lib=${1:-sound-event-detection_amd/sed_amd/libsedhip.so}
for args in "" "-H 440" "-P 0" "-P 0 -H 440" "-M 64" "-P 0 -H 440 -M 64"; do
  echo "== kaldi44 $args"
  timeout -k 10 120 $b/kbench -c kaldi44 -n 50 $args $lib || exit 1
done
