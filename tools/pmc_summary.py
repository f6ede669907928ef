#!/usr/bin/env python3
"""Turn the rocprofv3 counter CSVs of two PMC passes over bench.py into the traffic summary that
bench.py quotes in `roofline.traffic`.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv \
        -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv \
        -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline
    python3 tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r02_pmc_traffic.json

Corrections are the ones MI355X_MICROARCH.md "HBM" prescribes: the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a 16-B/lane coalesced streaming read (x2);
WRITE_SIZE is exact for 16-B/lane stores.  Separate passes, --kernel-trace only.
"""
import csv
import glob
import json
import os
import sys


def read_counter(d, counter, kernel_substr):
    vals = []
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                    vals.append((int(row["Dispatch_Id"]), float(row["Counter_Value"]),
                                 int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                 row["Kernel_Name"]))
    vals.sort()
    return vals


def sixteenths(vals):
    n = len(vals)
    return [round(sum(v[1] for v in vals[n * i // 16:n * (i + 1) // 16]) /
                  max(1, n * (i + 1) // 16 - n * i // 16), 1) for i in range(16)]


def main():
    fdir, wdir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "relax_k<float"
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 16384
    fetch = read_counter(fdir, "FETCH_SIZE", kernel)
    write = read_counter(wdir, "WRITE_SIZE", kernel)
    if not fetch or not write:
        raise SystemExit("no %s dispatches with FETCH_SIZE / WRITE_SIZE found" % kernel)
    f_mean = sum(v[1] for v in fetch) / len(fetch)
    w_mean = sum(v[1] for v in write) / len(write)
    fetch_b = f_mean * 1024.0 * 2.0
    write_b = w_mean * 1024.0
    alg_read = 4.0 * n * n
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), "
                  "ROCm 7.2, one MI355X, over the benchmark command itself: python3 bench.py --steps 1 "
                  "--warmup 0 --no-extras --no-cpu-baseline (torch-free, launches throttled to 512 in "
                  "flight); summarised by tools/pmc_summary.py",
        "kernel": fetch[0][3].split("(")[0],
        "launches_fetch_pass": len(fetch), "launches_write_pass": len(write),
        "FETCH_SIZE_KiB_mean_raw": f_mean, "WRITE_SIZE_KiB_mean_raw": w_mean,
        "correction": "MI355X_MICROARCH.md 'HBM': counters are in KiB; on gfx950 FETCH_SIZE reports "
                      "exactly 1/2 of the bytes of a 16-B/lane coalesced streaming read -> x2; "
                      "WRITE_SIZE is exact for 16-B/lane stores",
        "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": fetch_b + write_b,
        "algorithmic_read_bytes_per_launch": alg_read,
        "fetch_over_algorithmic": fetch_b / alg_read,
        "avg_kernel_ns_under_pmc_fetch_pass": sum(v[2] for v in fetch) / len(fetch),
        "by_k_sixteenth_fetch_KiB_raw": sixteenths(fetch),
        "by_k_sixteenth_write_KiB_raw": sixteenths(write),
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("launches_fetch_pass", "fetch_bytes_per_launch",
                                          "write_bytes_per_launch", "fetch_over_algorithmic")}))


if __name__ == "__main__":
    main()
