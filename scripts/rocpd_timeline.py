#!/usr/bin/env python3
"""Timeline of the last dispatches in a rocprofv3 rocpd database: start and end of every kernel relative to the first one listed.
usage: rocpd_timeline.py <db> [count]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns:", cols)
rows = db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, stream_id, queue_id from kernels order by start").fetchall() \
    if "stream_id" in cols and "queue_id" in cols else \
    [r + (None, None) for r in db.execute("select name, start, end, grid_x*grid_y*grid_z, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start")]
rows = rows[-count:]
t0 = rows[0][1]
for name, s, e, grid, wg, vg, ag, lds, st, q in rows:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name).replace("nmfx::", "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  blocks {grid // max(wg, 1):5d} x {wg:4d}  vgpr {vg}+{ag} lds {lds}  q {q} s {st}  {name[:90]}")
