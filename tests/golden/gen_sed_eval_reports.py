#!/usr/bin/env python3
"""Copy the sed_eval reports the reference printed into a fixture.

Runs ONLY in the build container (needs /root/reference).  notebooks/evaluation_test.ipynb holds, as cell outputs, the
text `print(event_based_metrics); print(segment_based_metrics)` produced (main.py:46-47) for six prediction files with
t_collar = 0.5 and segments of 1 s.  The predictions and the ground truth behind them are not in the reference, so the
reports cannot be recomputed; what they do pin is sed_eval's arithmetic from counts to rates and its report layout.
Output: sed_eval_reports.json = {prediction file: report text}, read by tests/test_metrics.py.
"""
import json
import os
import re

NB = "/root/reference/notebooks/evaluation_test.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sed_eval_reports.json")

nb = json.load(open(NB))
reports = {}
for cell in nb["cells"]:
    src = "".join(cell.get("source", []))
    m = re.search(r"open\('outputs/(\w+)\.json'", src)
    if not m or "calculate_metrics(" not in src:
        continue
    text = "".join("".join(o.get("text", [])) for o in cell.get("outputs", []) if o.get("output_type") == "stream")
    assert text.startswith("Event based metrics") and "Segment based metrics" in text, m.group(1)
    reports[m.group(1)] = text
assert len(reports) == 6, sorted(reports)
json.dump(reports, open(OUT, "w"), indent=1)
print("wrote", OUT, {k: len(v) for k, v in reports.items()})
