#!/usr/bin/env python3
"""Exact `_path` lists at the headline size: fused engine with the path trace, then sampled queries."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or (4096, 16384):
    rate, nxt = synth.d1_uniform(n, np.float32, 20243)
    with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
        dm.enable_path_log()
        for rep in range(2):
            dm.upload(rate, nxt)
            t0 = time.perf_counter()
            dm.solve()
            t = time.perf_counter() - t0
        print("N=%d f32 + next + path trace, fused engine: %.1f ms (%.3e relaxations/s)" % (n, 1e3 * t, n ** 3 / t), flush=True)
        rnd = np.random.default_rng(1)
        _, nx, _ = dm.download()
        t0 = time.perf_counter()
        lens, same = [], 0
        for _ in range(300):
            i, j = (int(x) for x in rnd.integers(0, n, 2))
            p = dm.query_exact(i, j)[1]
            lens.append(len(p))
            same += p == (engine.follow_path(nx, i, j) if i != j else [])
        print("  300 exact queries: %.2f ms each incl. the host-side walk, mean length %.1f, max %d, %d equal to the next-hop walk"
              % (1e3 * (time.perf_counter() - t0) / 300, np.mean(lens), max(lens), same), flush=True)
