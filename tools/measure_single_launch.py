#!/usr/bin/env python3
"""Latency of the single-launch solve (small_solve, n <= 128) against the per-k engine on the same
matrices: device-resident f64 + next + hops (what the host mirror solves), restore-from-pristine
cost subtracted; and of a solve with the path trace through fwx_matrix_solve (upload time subtracted)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

dev = torch.device("cuda:0")
REPS = 20


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS


for n in (4, 16, 48, 64, 65, 96, 120, 128):
    rate, nxt, hops = synth.make("d2", n, np.float64, seed=3)
    r0, n0, h0 = (torch.from_numpy(a).to(dev) for a in (rate, nxt, hops))
    r, nx, hp = r0.clone(), n0.clone(), h0.clone()

    def restore():
        r.copy_(r0)
        nx.copy_(n0)
        hp.copy_(h0)
        torch.cuda.synchronize()

    t_restore = timed(restore)
    line = "n=%4d f64+next+hops:" % n
    for name, code in (("per-k", engine.FWX_ENGINE_PERK), ("single launch", engine.FWX_ENGINE_AUTO)):
        def run():
            restore()
            engine.dev_solve(r, next_t=nx, hops_t=hp, engine=code)
        line += "  %s %.1f us" % (name, 1e6 * (timed(run) - t_restore))
    with engine.DeviceMatrix(n, np.float64, with_next=True, with_hops=True) as dm:
        dm.enable_path_log()
        t_up = timed(lambda: dm.upload(rate, nxt, hops))

        def traced(code):
            dm.upload(rate, nxt, hops)
            dm.solve(engine=code)
        line += "  traced solve %.1f us" % (1e6 * (timed(lambda: traced(engine.FWX_ENGINE_AUTO)) - t_up))
        line += "  [per-k traced %.1f us]" % (1e6 * (timed(lambda: traced(engine.FWX_ENGINE_PERK)) - t_up))
    print(line, flush=True)
