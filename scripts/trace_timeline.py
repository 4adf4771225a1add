"""Timeline of one solver iteration from a rocprofv3 rocpd database: every kernel with its start offset, duration and queue, so that
launches of two streams that (should) overlap can be seen side by side.  usage: trace_timeline.py <db> [anchor kernel substring] [count]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "gemm_mfma"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
cols = [c[1] for c in db.execute(f"pragma table_info({kt[0]})")]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
sel = "name,start,end" + (f",{qcol}" if qcol else "")
rows = db.execute(f"select {sel} from {kt[0]} order by start").fetchall()
rows = [r for r in rows if "nmfx" in r[0]]
mid = len(rows) // 2
t0 = rows[mid][1]
for r in rows[mid:mid + count]:
    n = re.sub(r"void nmfx::|\(.*", "", r[0])
    n = re.sub(r"<.*", "", n)
    print(f"{(r[1]-t0)/1e3:9.2f} us  +{(r[2]-r[1])/1e3:8.2f} us  q={r[3] if qcol else '-'}  {n}")
