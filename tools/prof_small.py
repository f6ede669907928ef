#!/usr/bin/env python3
"""Kernel-level timing of the fused engine at small N (run under rocprofv3 --kernel-trace --stats)."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
rate, nxt, _ = synth.make("d1", n, dt, seed=1)
dev = torch.device("cuda:0")
r0 = torch.from_numpy(rate).to(dev)
r = r0.clone()
for _ in range(5):
    r.copy_(r0)
    engine.dev_solve(r, engine=engine.FWX_ENGINE_FUSED)
torch.cuda.synchronize()
