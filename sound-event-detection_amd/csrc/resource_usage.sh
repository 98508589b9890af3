#!/bin/bash
# Prints one line per kernel of a .hip file: registers, spills, scratch, occupancy (hipcc resource remarks).
# usage: resource_usage.sh [file.hip] [extra hipcc flags]
f=${1:-sed_frontend.hip}
shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include "$@" -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur=t.split(":",1)[1].strip()[:70].ljust(72)
    else:
        k,v=t.split(":",1)
        k=k.strip()
        if k in ("VGPRs","AGPRs","TotalSGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","VGPRs Spill","SGPRs Spill","LDS Size [bytes/block]"):
            cur+=f" {k.split()[0]}={v.strip()}"
if cur: print(cur)
'
