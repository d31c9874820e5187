#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, launches, average duration and the average value of every counter
per launch.  usage: pmc_summary.py <tag> <dir>...   (writes a table to stdout and gpurun_out/<tag>_pmc_summary.json)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, dirs = sys.argv[1], sys.argv[2:]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, dispatches]
dur = defaultdict(lambda: [0.0, 0])
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                agg[k][row["Counter_Name"]][0] += float(row["Counter_Value"])
                agg[k][row["Counter_Name"]][1] += 1
                key = (path, row["Dispatch_Id"])
                if key not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    seen.add(key)
                    dur[k][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
                    dur[k][1] += 1
out = {}
for k, cs in agg.items():
    out[k] = {c: v[0] / v[1] for c, v in cs.items() if v[1]}
    out[k]["launches_per_pass"] = max(v[1] for v in cs.values())
    if dur[k][1]:
        out[k]["avg_ms_profiled"] = dur[k][0] / dur[k][1]
counters = sorted({c for v in out.values() for c in v})
rows = sorted(out.items(), key=lambda kv: -kv[1].get("avg_ms_profiled", 0) * kv[1]["launches_per_pass"])
print("# per-launch averages; FETCH_SIZE / WRITE_SIZE in KB as reported by rocprofv3; SQ_* cycle counters in quad-cycles")
print("kernel | " + " | ".join(counters))
for k, v in rows:
    print(k[:70] + " | " + " | ".join(f"{v[c]:.4g}" if c in v else "-" for c in counters))
os.makedirs("gpurun_out", exist_ok=True)
with open(os.path.join("gpurun_out", f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
