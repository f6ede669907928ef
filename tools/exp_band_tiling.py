#!/usr/bin/env python3
"""Experiment: per-k kernel with a time-tiled band schedule.  For each panel of T pivots (snapshot
panel W from fwx_dev_panel_snap) sweep the matrix band by band, applying all T pivots to a band
(one relax_k launch per pivot and band) before moving on, so that launches 2..T of a band are
served from the 256 MiB Infinity Cache.  Same kernel, same bits; only the launch order changes."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import engine, synth  # noqa: E402

n = 16384
dev = torch.device("cuda:0")
rate_h, _ = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
r0 = torch.from_numpy(rate_h).to(dev)
ws = engine.FusedWorkspace(n, n, torch.float32, dev)

ref = r0.clone()
engine.dev_relax(ref, n, 0, 0, 2048)
torch.cuda.synchronize()


def run(T, band_rows, kmax=2048):
    r = r0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k0 in range(0, kmax, T):
        w = ws.w[0][:T]
        engine.dev_panel_snap(r[k0:k0 + T], n, k0, w)
        for b0 in range(0, n, band_rows):
            engine.dev_relax(r[b0:b0 + band_rows], n, b0, k0, k0 + T, pivots_t=w)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(r, ref))
    return dt, ok


def base(kmax=2048):
    r = r0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.dev_relax(r, n, 0, 0, kmax)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


base()
print("baseline per-k (whole matrix per launch): %.1f us per pivot" % (1e6 * base() / 2048))
for T in (4, 8, 16, 32, 64):
    for band in (1024, 2048, 3072):
        dt, ok = run(T, band)
        print("T=%2d band=%4d rows (%3d MiB): %.1f us per pivot  bit-identical=%s" % (
            T, band, band * n * 4 >> 20, 1e6 * dt / 2048, ok), flush=True)
