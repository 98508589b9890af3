#!/bin/bash
# Prints one markdown table row per kernel of a .hip file: registers, spills, scratch, occupancy (hipcc resource remarks).
# usage: resource_usage.sh [file.hip] [extra hipcc flags]
f=${1:-sed_frontend.hip}
shift
# the decoders and the resampler are built without fp contraction, as in the Makefile
case "$f" in *sed_frontend.hip) nc= ;; *) nc=-ffp-contract=off ;; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $nc "$@" -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, subprocess, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    k, v = t.split(":", 1)
    k, v = k.strip(), v.strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
print("| kernel | VGPRs | AGPRs | SGPRs | VGPR spills | SGPR spills | scratch B/lane | waves/SIMD |")
print("|---|---|---|---|---|---|---|---|")
for r, n in zip(rows, names):
    n = n.replace("sed::", "").replace("void ", "")
    n = re.sub(r"\((anonymous namespace::)?\w*Params\)$", "", n)
    print("| `%s` | %s | %s | %s | %s | %s | %s | %s |" % (n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
          r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
'
