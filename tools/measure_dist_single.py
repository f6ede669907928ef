#!/usr/bin/env python3
"""floydwarshall_amd.dist.solve_partitioned at world size 1 (no process group): time of the panel /
look-ahead schedule itself on one GPU, both engines, against the plain single-GPU solve."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import dist as fwdist, engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rate, nxt = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
dev = torch.device("cuda:0")
pristine = torch.from_numpy(rate).to(dev)
out = {"n": n}
for name in ("perk", "fused"):
    backend = fwdist.HipBackend(name)
    ts = []
    for _ in range(3):
        r = pristine.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fwdist.solve_partitioned(r, n, 0, 1, backend=backend)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out["dist_" + name + "_ms"] = round(1e3 * min(ts), 2)
for name, eng in (("perk", engine.FWX_ENGINE_PERK), ("fused", engine.FWX_ENGINE_FUSED)):
    ts = []
    for _ in range(3):
        r = pristine.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.dev_solve(r, engine=eng)
        ts.append(time.perf_counter() - t0)
    out["single_" + name + "_ms"] = round(1e3 * min(ts), 2)
print(json.dumps(out))
