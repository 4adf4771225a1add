"""Summarise rocprofv3 (rocpd sqlite) output into the text files committed under profiles/.

usage: rocpd_summary.py <stats.db> [--pmc FETCH=<db> WRITE=<db>] > profiles/rNN_....md
Kernels are grouped by (name, grid size) because one GEMM template instantiation serves several call sites
(e.g. the 2*p*n*k GEMM and the k x k Gram use the same EpiStore kernel with different grids).
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("nmfx::", "")


def kernel_table(db):
    rows = db.execute("select name, grid_x*grid_y*grid_z, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, duration "
                      "from kernels").fetchall()
    agg = defaultdict(list)
    meta = {}
    for name, grid, wg, vg, ag, lds, dur in rows:
        key = (short(name), grid)
        agg[key].append(dur)
        meta[key] = (wg, vg, ag, lds)
    total = sum(sum(v) for v in agg.values())
    out = []
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        wg, vg, ag, lds = meta[key]
        out.append((key[0], key[1] // max(wg, 1), wg, vg, ag, lds, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3,
                    max(v) / 1e3, 100.0 * sum(v) / total))
    return out


def pmc_table(db, counter):
    rows = db.execute("select kernel_name, grid_size, value from counters_collection where counter_name=?", (counter,)).fetchall()
    agg = defaultdict(list)
    for name, grid, val in rows:
        agg[(short(name), grid)].append(val)
    return agg


def main():
    stats = sys.argv[1]
    db = sqlite3.connect(stats)
    print("## rocprofv3 --kernel-trace --stats (grouped by kernel and grid)\n")
    print("| kernel | blocks | wg | vgpr | agpr | lds B | calls | total us | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in kernel_table(db):
        if r[-1] < 0.05:
            continue
        print(f"| `{r[0][:110]}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]:.1f} | {r[8]:.2f} | {r[9]:.2f} | {r[10]:.2f} | {r[11]:.2f} |")
    for arg in sys.argv[2:]:
        if "=" not in arg:
            continue
        ctr, path = arg.split("=", 1)
        dbp = sqlite3.connect(path)
        names = [r[0] for r in dbp.execute("select distinct counter_name from counters_collection")]
        for cname in names:
            print(f"\n## PMC {cname} per dispatch (rocprofv3 --pmc {cname}), raw counter values (FETCH_SIZE/WRITE_SIZE are KiB)\n")
            print("| kernel | grid threads | dispatches | mean | min | max |")
            print("|---|---|---|---|---|---|")
            agg = pmc_table(dbp, cname)
            for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                if "nmfx" not in key[0] and "gemm" not in key[0] and "kernel" not in key[0]:
                    continue
                if "at::native" in key[0]:
                    continue
                print(f"| `{key[0][:110]}` | {key[1]} | {len(v)} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")


if __name__ == "__main__":
    main()
