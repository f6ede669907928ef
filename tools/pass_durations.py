#!/usr/bin/env python3
"""Per-pass kernel durations from a rocprofv3 --kernel-trace CSV: for each kernel name prefix, the
launch durations in dispatch order, averaged per sixteenth of the launches (how the cost of a
64-pivot pass moves through the solve).  usage: pass_durations.py <dir> [substr ...]"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    subs = sys.argv[2:] or ["fused_main"]
    rows = []
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    for sub in subs:
        sel = [(e - s) for s, e, k in rows if sub in k]
        if not sel:
            print(json.dumps({"kernel": sub, "launches": 0}))
            continue
        n = len(sel)
        six = [round(sum(sel[n * i // 16:n * (i + 1) // 16]) / max(1, n * (i + 1) // 16 - n * i // 16) / 1e3, 1)
               for i in range(16)]
        print(json.dumps({"kernel": sub, "launches": n, "total_ms": round(sum(sel) / 1e6, 2),
                          "avg_us": round(sum(sel) / n / 1e3, 1), "avg_us_by_sixteenth": six}))


if __name__ == "__main__":
    main()
