#!/usr/bin/env python3
"""Mean PMC counter values per kernel from a rocprofv3 --pmc run (CSV output).
usage: pmc_summary.py <dir with *counter_collection.csv> <kernel name substring>"""
import csv
import glob
import sys
from collections import defaultdict

root, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        e = acc[(r["Kernel_Name"][:40], r["Counter_Name"])]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
for (k, c), e in sorted(acc.items()):
    print(f"{k:42s} {c:28s} n={e[0]:5d} mean={e[1] / e[0]:16.1f}")
