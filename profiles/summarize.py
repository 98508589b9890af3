#!/usr/bin/env python3
"""Summarise a profiles/collect.sh run: per-kernel time (rocprofv3 --kernel-trace --stats) and the mean per-dispatch
PMC counters of the kernels named sed::*.  Usage: python profiles/summarize.py gpurun_out/prof_<tag> profiles/<tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:90]


def main(src, dst):
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    out = {"kernels": [], "pmc": {}}
    for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "sed::" in row["Name"] or float(row["Percentage"]) > 1.0:
                out["kernels"].append({"name": short(row["Name"]), "calls": int(row["Calls"]),
                                       "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3,
                                       "max_us": float(row["MaxNs"]) / 1e3, "pct": float(row["Percentage"])})
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "sed::" not in row["Kernel_Name"]:
                continue
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                       "lds_bytes": int(row["LDS_Block_Size"]), "vgpr": int(row["VGPR_Count"]),
                       "agpr": int(row["Accum_VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                       "scratch": int(row["Scratch_Size"])}
    for k, counters in acc.items():
        out["pmc"][k] = {"launch": meta[k], "mean_per_dispatch": {c: sum(v) / len(v) for c, v in sorted(counters.items())},
                         "dispatches": {c: len(v) for c, v in counters.items()}}
    json.dump(out, open(dst + ".json", "w"), indent=1)
    with open(dst + ".md", "w") as md:
        md.write(f"# rocprofv3 summary ({os.path.basename(src)})\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for k in sorted(out["kernels"], key=lambda r: -r["pct"]):
            md.write(f"| `{k['name']}` | {k['calls']} | {k['avg_us']:.1f} | {k['min_us']:.1f} | {k['max_us']:.1f} | {k['pct']:.2f} |\n")
        for k, v in out["pmc"].items():
            md.write(f"\n## PMC, mean per dispatch: `{k}`\n\nlaunch: {v['launch']}\n\n| counter | value |\n|---|---|\n")
            for c, val in v["mean_per_dispatch"].items():
                md.write(f"| {c} | {val:.4g} |\n")
    print(open(dst + ".md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
