#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd SQLite database (the default
output format of ROCm 7.2): calls, total / average / min / max duration, grid.  With --timeline N the
first N dispatches after --skip are listed in start order (who ran beside whom).
usage: rocpd_summary.py results.db [--timeline N] [--skip M]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("""select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.queue_id
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                      order by d.start""").fetchall()


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("fwx::(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


agg = {}
for name, st, en, gx, gy, wx, q in rows:
    k = (short(name), gx // max(wx, 1), gy)
    a = agg.setdefault(k, [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += en - st
    a[2] = min(a[2], en - st)
    a[3] = max(a[3], en - st)
tot = sum(a[1] for a in agg.values())
print("%-72s %10s %8s %10s %9s %9s %9s %6s" % ("kernel", "grid", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %10s %8d %10.2f %9.1f %9.1f %9.1f %6.1f" % (k[0], "%dx%d" % (k[1], k[2]), a[0], a[1] / 1e6,
                                                        a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
if "--timeline" in sys.argv:
    n = int(sys.argv[sys.argv.index("--timeline") + 1])
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    t0 = rows[skip][1]
    for name, st, en, gx, gy, wx, q in rows[skip:skip + n]:
        print("%9.1f us  +%8.1f us  q%-3d %-50s %dx%d" % ((st - t0) / 1e3, (en - st) / 1e3, q, short(name)[:50],
                                                         gx // max(wx, 1), gy))
