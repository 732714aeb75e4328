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
# optional: tools/pmc_summary.py <dir> <out.json> <traffic.json> <workload key>  -> merges the HBM traffic of
# this profile into the table bench.py reads
if len(sys.argv) >= 5 and "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    path, key = sys.argv[3], sys.argv[4]
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[key] = {"FETCH_SIZE_KiB": out["FETCH_SIZE"]["mean_per_dispatch"], "WRITE_SIZE_KiB": out["WRITE_SIZE"]["mean_per_dispatch"],
                  "source": os.path.basename(sys.argv[2])}
    for name in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_REQ_sum", "TCC_HIT_sum"):
        if name in out:
            table[key][name] = out[name]["mean_per_dispatch"]
    # which kernels the counters were taken from: the hash of the kernel sources this tree's library was built from
    # (bench.py drops an entry whose hash is not its own library's)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import ray_tracer_amd
    table[key]["kernel_build"] = ray_tracer_amd.kernel_build()
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
