#!/usr/bin/env python3
"""Latency of the per-call host API (fwx_solve_f64 / _f32: upload, solve, download, blocking) for the
reference's own matrix sizes and a little above: f64 + next + hops, median of 200 calls."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from floydwarshall_amd import engine, synth  # noqa: E402

for dt in (np.float64, np.float32):
    for n in (4, 16, 64, 128, 160, 256, 512, 1024):
        rate, nxt, hops = synth.make("d1", n, dt, seed=5)
        ts = []
        for it in range(60 if n >= 512 else 200):
            r, x, h = rate.copy(), nxt.copy(), hops.copy()
            t0 = time.perf_counter()
            engine.solve(r, x, h)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[5:])
        print("%s n=%4d  median %.3f ms  min %.3f ms" % (np.dtype(dt).name, n, 1e3 * ts[len(ts) // 2], 1e3 * ts[0]),
              flush=True)
