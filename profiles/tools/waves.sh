#!/bin/bash
export SED_HIP_LIB=$PWD/sound-event-detection_amd/sed_amd/libsedhip_ablate.so
for w in 2 4 6 8 10 12; do
  SED_DEBUG_WAVES=$w timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('waves', $w, 'kernel_us', round(r['roofline']['kernel_ms']*1e3,1))" || exit 1
done
