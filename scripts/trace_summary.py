#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel, launches that did work (longer than --min-us) and launches that were
device-side no-ops (gated kernels of the speculative enqueue), with their mean durations.
usage: trace_summary.py <dir with *_kernel_trace.csv> [--min-us 8]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 8.0
rows = defaultdict(lambda: [0, 0.0, 0, 0.0])
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"]
        name = name[:150]
        e = rows[name]
        if us >= min_us:
            e[0] += 1; e[1] += us
        else:
            e[2] += 1; e[3] += us
tot = sum(e[1] + e[3] for e in rows.values())
print(f"total kernel time {tot / 1e3:.2f} ms")
for name, e in sorted(rows.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    w = f"{e[0]:6d} x {e[1] / max(e[0], 1):9.2f} us" if e[0] else " " * 23
    i = f"{e[2]:6d} x {e[3] / max(e[2], 1):6.2f} us (short)" if e[2] else ""
    print(f"{(e[1] + e[3]) / tot * 100:5.1f}%  {w}  {i}  {name}")
