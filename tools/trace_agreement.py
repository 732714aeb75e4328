#!/usr/bin/env python3
"""Per-launch durations of the timed render kernel in a rocprofv3 kernel trace beside the figures of the bench line
that ran under that trace.

usage: tools/trace_agreement.py <kernel_trace.csv> <bench line .json> "<title>"   (prints one section)
"""
import csv
import json
import sys


def main():
    trace, line, title = sys.argv[1:4]
    want = "render_kernel<true, false, false>"
    rows = []
    with open(trace, newline="") as f:
        for r in csv.DictReader(f):
            if want in r["Kernel_Name"].replace("(bool)1", "true").replace("(bool)0", "false"):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    ms = [(e - s) / 1e6 for s, e in rows]
    d = json.load(open(line))
    cfg = d["roofline"]
    steps, warmup = d["steps"], d["warmup"]
    print(f"## {title}: {len(ms)} launches of the timed kernel; JSON: kernel_ms {cfg.get('kernel_ms')}, "
          f"kernel_ms_by_stream_events {cfg.get('kernel_ms_by_stream_events')}, ms_per_step {d['ms_per_step']}")
    first = 1 + warmup           # one census launch of the timed mode, then the warm-up
    timed = ms[first:first + steps]
    if timed:
        print(f"launches {first}..{first + steps - 1} (the {steps} timed frames of the headline family): "
              f"mean {sum(timed) / len(timed):.3f} ms, min {min(timed):.3f}, max {max(timed):.3f}")
    print("all launches, ms: " + " ".join(f"{x:.1f}" for x in ms))
    print()


if __name__ == "__main__":
    main()
