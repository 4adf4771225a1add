"""Summarise kernel durations and inter-kernel gaps of the solver loop from a rocprofv3 rocpd database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
rows = [(re.sub(r'void nmfx::|<.*|\(.*', '', n), s, e) for n, s, e in rows if 'nmfx' in n]
lo, hi = len(rows) // 3, 2 * len(rows) // 3
gaps = {}
tot_gap = tot_k = 0
for (n0, s0, e0), (n1, s1, e1) in zip(rows[lo:hi], rows[lo + 1:hi + 1]):
    g = (s1 - e0) / 1e3
    gaps.setdefault((n0, n1), []).append(g)
    tot_gap += g
    tot_k += (e0 - s0) / 1e3
for k, v in gaps.items():
    print(f"{k[0]:28s}->{k[1]:28s} n={len(v):3d} gap avg {sum(v)/len(v):6.2f} us")
print(f"total gap {tot_gap:.0f} us, total kernel {tot_k:.0f} us, span {(rows[hi][1]-rows[lo][1])/1e3:.0f} us")
