#!/usr/bin/env python3
"""Fused engine in the reference's precision (f64), rates only and with the next-hop matrix."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or (4096, 8192, 16384):
    rate, nxt = synth.d1_uniform(n, np.float64, 5)
    r0, n0 = torch.from_numpy(rate).to(dev), torch.from_numpy(nxt).to(dev)
    del rate, nxt
    r, nx = torch.empty_like(r0), torch.empty_like(n0)
    for with_next in (False, True):
        def run():
            r.copy_(r0)
            if with_next:
                nx.copy_(n0)
            engine.dev_solve(r, next_t=nx if with_next else None, engine=engine.FWX_ENGINE_FUSED)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("N=%5d f64 fused %s: %.1f ms = %.3e relaxations/s, %.0f us per 64-pivot pass"
              % (n, "rate+next" if with_next else "rates only", 1e3 * dt, n ** 3 / dt, 1e6 * dt / (n / 64)),
              flush=True)
