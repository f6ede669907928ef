#!/usr/bin/env python3
"""Per-kernel means of one PMC counter from a rocprofv3 --pmc CSV directory.
usage: pmc_by_kernel.py DIR COUNTER  -> JSON lines {kernel, launches, mean, sum, mean_ns}"""
import csv
import glob
import json
import os
import re
import sys


def main():
    d, counter = sys.argv[1], sys.argv[2]
    acc = {}
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                k = re.sub(r"^void ", "", k).split("(")[0]
                a = acc.setdefault(k, [0, 0.0, 0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    for k, (c, s, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(json.dumps({"kernel": k[:110], "counter": counter, "launches": c, "mean": round(s / c, 1),
                          "sum": round(s, 1), "mean_ns": round(ns / c)}))


if __name__ == "__main__":
    main()
