"""Summarises a rocprofv3 kernel trace of tools/build_time.py: per-kernel totals and the
per-level durations of the last build."""
import csv
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_build"
rows = list(csv.DictReader(open(f"{d}/build_kernel_stats.csv")))
for r in rows:
    nm = r["Name"].split("(")[0].replace("rtx::", "").replace("anonymous namespace)::", "")
    print(f"{nm[-30:]:30s} {r['Calls']:>5s} {int(r['TotalDurationNs'])/1e6:8.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  max {int(r['MaxNs'])/1e3:6.0f} us")
tr = sorted(csv.DictReader(open(f"{d}/build_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
ci = [i for i, r in enumerate(tr) if "centroid_kernel" in r["Kernel_Name"]][-1]
seg = tr[ci:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for name in ("bounds_kernel", "bin_kernel", "plane_kernel", "flag_kernel", "split_kernel", "holes_kernel", "permute_kernel", "subtree_kernel", "emit_kernel"):
    print(name, [round(dur(r)) for r in seg if name in r["Kernel_Name"]])
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
busy = sum(dur(r) for r in seg) / 1e3
print(f"last build: span {span:.2f} ms, kernels busy {busy:.2f} ms, launches {len(seg)}")
