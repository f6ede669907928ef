#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the headline bench): config 2 (N=1024 fp64),
end-to-end through the host-buffer C ABI (PCIe inclusive), f64 engines.  One MI355X."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
dev = torch.device("cuda:0")
for n, dt, name in [(1024, np.float64, "config2_n1024_f64"), (2048, np.float64, "n2048_f64"),
                    (4096, np.float64, "n4096_f64"), (1024, np.float32, "n1024_f32")]:
    rate, nxt, _ = synth.make("d1", n, dt, seed=synth.BASE_SEED + 1)
    r0 = torch.from_numpy(rate).to(dev)
    n0 = torch.from_numpy(nxt).to(dev)
    r, nx = r0.clone(), n0.clone()
    res = {}
    for eng, code in (("perk", engine.FWX_ENGINE_PERK), ("fused", engine.FWX_ENGINE_FUSED)):
        for with_next in (False, True):
            def run():
                r.copy_(r0)
                nx.copy_(n0)
                engine.dev_solve(r, next_t=nx if with_next else None, engine=code)
            t = timed(run, 5)
            res["%s%s" % (eng, "+next" if with_next else "")] = {
                "ms": 1e3 * t, "relax_per_s": n ** 3 / t}
    out[name] = res

# end-to-end through fwx_solve_f32 (host buffers: H2D + solve + D2H), N=16384
n = 16384
rate, _ = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
for eng, code in (("fused", engine.FWX_ENGINE_FUSED), ("perk", engine.FWX_ENGINE_PERK)):
    work = rate.copy()
    t0 = time.perf_counter()
    engine.solve(work, engine=code)
    out["e2e_host_buffers_n16384_f32_" + eng] = {"ms": 1e3 * (time.perf_counter() - t0)}
print(json.dumps(out, indent=1))
