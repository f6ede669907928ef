#!/usr/bin/env python3
"""Overhead of the partitioned schedule itself: the same N x N solve through a multi handle with
P LOGICAL partitions of ONE GPU (fwx_matrix_create_multi, device 0 listed P times, peer-copy
exchange) against the single-device handle.  Not a scaling number: all partitions share one GPU."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from floydwarshall_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
with_next = "--next" in sys.argv
rate, nxt = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
ref = None
for parts in (0, 1, 2, 4, 8):
    h = engine.DeviceMatrix(n, np.float32, with_next=with_next, device=0) if parts == 0 else \
        engine.DeviceMatrix(n, np.float32, with_next=with_next, devices=[0] * parts)
    ts = []
    for _ in range(3):
        h.upload(rate, nxt if with_next else None)
        t0 = time.perf_counter()
        h.solve()
        ts.append(time.perf_counter() - t0)
    timing = None
    if parts:
        # one more solve with per-step event spans (round 4): chain vs bulk per 128-pivot launch
        h.set_timing(True)
        h.upload(rate, nxt if with_next else None)
        h.solve()
        timing = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in h.timing().items()}
        h.set_timing(False)
    r = h.download()[0]
    if ref is None:
        ref = r
        single_ms = 1e3 * min(ts)
    print(json.dumps({"n": n, "next": with_next, "partitions": parts or "single-device handle",
                      "best_ms": round(1e3 * min(ts), 2), "ms": [round(1e3 * t, 2) for t in ts],
                      "overhead_over_single_device": round(1e3 * min(ts) / single_ms - 1.0, 4),
                      "timing": timing,
                      "bits_equal_single": bool(np.array_equal(r.view(np.uint32), ref.view(np.uint32)))}),
          flush=True)
    h.close()
