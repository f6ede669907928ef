#!/usr/bin/env python3
"""Per-k vs fused vs small_solve latency for small matrices (device-resident, f64 + next + no hops)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

dev = torch.device("cuda:0")
for n in (32, 48, 64, 72, 80, 96, 112, 128, 132, 160, 192, 224, 256, 384, 512, 768, 1024):
    for dt in (np.float64, np.float32):
        rate, nxt, _ = synth.make("d1", n, dt, seed=3)
        r0, n0 = torch.from_numpy(rate).to(dev), torch.from_numpy(nxt).to(dev)
        r, nx = r0.clone(), n0.clone()
        line = "n=%4d %s" % (n, dt.__name__)
        for name, code in (("perk", engine.FWX_ENGINE_PERK), ("fused", engine.FWX_ENGINE_FUSED),
                           ("auto", engine.FWX_ENGINE_AUTO)):
            def run():
                r.copy_(r0)
                nx.copy_(n0)
                engine.dev_solve(r, next_t=nx, engine=code)
            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            line += "  %s %.3f ms" % (name, 1e2 * (time.perf_counter() - t0))
        print(line, flush=True)
