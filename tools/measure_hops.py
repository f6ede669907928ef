#!/usr/bin/env python3
"""fwx_solve_f32 with rate + next + hops (host buffers, PCIe included): fused engine + path trace +
length reconstruction against the per-k engine."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

engine.solve(np.ones((8, 8), dtype=np.float32))          # runtime start-up out of the way
for n in (1024, 4096, 16384):
    rate, nxt = synth.d1_uniform(n, np.float32, 5)
    hops = (nxt >= 0).astype(np.int32)
    for name, code in (("fused + trace + lengths", engine.FWX_ENGINE_AUTO), ("per-k", engine.FWX_ENGINE_PERK)):
        if n == 16384 and code == engine.FWX_ENGINE_PERK:
            continue
        r, nx, hp = rate.copy(), nxt.copy(), hops.copy()
        t0 = time.perf_counter()
        engine.solve(r, nx, hp, engine=code)
        print("N=%d f32 rate+next+hops through fwx_solve_f32 (PCIe included), %s: %.0f ms" % (n, name, 1e3 * (time.perf_counter() - t0)), flush=True)
