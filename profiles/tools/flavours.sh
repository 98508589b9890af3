#!/bin/bash
# Times every front-end flavour with one build of the library: profiles/tools/flavours.sh <lib.so>
lib=${1:-sound-event-detection_amd/sed_amd/libsedhip.so}
for c in htsat htsat_i16 yamnet kaldi crnn dbmel htsat1h htsat44 kaldi44; do
  timeout -k 10 120 profiles/tools/build/kbench -c $c -n 100 $lib || exit 1
done
