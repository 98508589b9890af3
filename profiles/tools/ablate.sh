#!/bin/bash
# times the fused kernel with phases of the streaming loop switched off (diagnostic build)
export SED_HIP_LIB=$PWD/sound-event-detection_amd/sed_amd/libsedhip_ablate.so
for skip in ${SKIPS:-0 1 2 3 8 16 32 64 96 127 126}; do
  SED_DEBUG_SKIP=$skip timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('skip', $skip, 'kernel_us', round(r['roofline']['kernel_ms']*1e3,1))" || exit 1
done
