#!/usr/bin/env python3
"""Timeline of one solve from a rocprofv3 --kernel-trace CSV: per kernel family the launches' durations, and for
the main launches the gaps between one's end and the next one's start (what the side chain costs the sweep).
usage: timeline.py <kernel_trace.csv> [skip_solves]   (analyses the LAST solve: the launches after the last
nonneg_check / upload gap)"""
import csv
import sys
from collections import defaultdict


def family(name):
    for key in ("fused_main_arg_f64", "fused_main_arg", "fused_main_max_f64", "fused_main_max", "fused_main",
                "fused_panels_next_f32", "fused_panels", "fused_rowpanel", "fused_colpanel", "nonneg_check", "relax_k"):
        if key in name:
            return key
    return name[:40]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1)))
    rows.sort()
    # last solve = launches after the last domain check
    last = max(i for i, r in enumerate(rows) if "nonneg_check" in r[2])
    rows = rows[last + 1:]
    t0 = rows[0][0]
    fam = defaultdict(list)
    for s, e, name, gx, gy in rows:
        fam[(family(name), gx * gy)].append((s - t0, e - t0))
    print("solve span: %.1f us, %d launches" % ((max(r[1] for r in rows) - t0) / 1e3, len(rows)))
    for (k, g), v in sorted(fam.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        d = [e - s for s, e in v]
        print("  %-26s grid %9d  x%4d  avg %8.1f us  min %8.1f  max %8.1f  total %9.1f us" %
              (k, g, len(v), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(d) / 1e3))
    # the big main launches: biggest grid of a main family
    mains = [(k, g) for (k, g) in fam if k.startswith("fused_main")]
    if mains:
        big = max(mains, key=lambda kg: kg[1])
        v = sorted(fam[big])
        gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        dur = [e - s for s, e in v]
        print("main launches (%s grid %d): %d, avg %.1f us; gap between them avg %.1f us, max %.1f us; sum of durations %.1f us = %.3f of the span"
              % (big[0], big[1], len(v), sum(dur) / len(dur) / 1e3, sum(gaps) / max(len(gaps), 1) / 1e3,
                 max(gaps or [0]) / 1e3, sum(dur) / 1e3, sum(dur) / (max(r[1] for r in rows) - t0)))
        if "--dump" in sys.argv:
            for s, e, name, gx, gy in rows[:60]:
                print("    %9.1f %9.1f  %-24s %d" % ((s - t0) / 1e3, (e - t0) / 1e3, family(name), gx * gy))


if __name__ == "__main__":
    try:
        main()
    except BrokenPipeError:          # piped into `head`
        pass
