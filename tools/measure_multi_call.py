#!/usr/bin/env python3
"""Per-call cost of the one-shot partitioned entry point (fwx_solve_multi_f64: create / take from the
pool + upload + solve + download) against the same work on a handle that already exists.
usage: measure_multi_call.py [n [parts]]   (default 4096 2; logical partitions of device 0)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from floydwarshall_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rate, nxt = synth.d1_uniform(n, np.float64, synth.BASE_SEED + 7)
calls = []
for _ in range(5):
    r, x = rate.copy(), nxt.copy()
    t0 = time.perf_counter()
    engine.solve_multi(r, x, devices=[0] * parts)
    calls.append(time.perf_counter() - t0)
with engine.DeviceMatrix(n, np.float64, with_next=True, devices=[0] * parts) as dm:
    hs = []
    for _ in range(4):
        r, x = rate.copy(), nxt.copy()
        t0 = time.perf_counter()
        dm.upload(r, x)
        dm.solve()
        dm.download()
        hs.append(time.perf_counter() - t0)
print(json.dumps({"n": n, "dtype": "f64", "fields": "rate+next", "partitions": parts,
                  "one_shot_call_ms": [round(1e3 * t, 2) for t in calls],
                  "handle_upload_solve_download_ms": [round(1e3 * t, 2) for t in hs],
                  "second_call_over_handle": round(min(calls[1:]) / min(hs[1:]), 3),
                  "what": "first call creates the handle (slabs, streams, events); later calls take it "
                          "from the pool; handle = upload + solve + download on a live handle "
                          "(download into fresh arrays)"}))
