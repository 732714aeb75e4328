#!/usr/bin/env python3
"""Folds the CSVs of tools/pmc_passes.sh into one JSON: per-dispatch means of every counter for the
render kernel.  usage: tools/pmc_summary.py gpurun_out/pmc/<tag> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "render_kernel" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row["Name"]:
            out["kernel_trace"] = {"name": row["Name"], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
json.dump(out, sys.stdout if len(sys.argv) < 3 else open(sys.argv[2], "w"), indent=1, sort_keys=True)
