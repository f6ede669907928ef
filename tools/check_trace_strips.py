#!/usr/bin/env python3
"""One-off validation (slow: the python restatement is O(n^3)): the path trace written by relax_k
with TWO column strips (f64, n = 516 > 512), or by the fused engine (9 passes), on a tie-heavy
input, every sampled list against the list-faithful restatement of Algorithms.hs:42-61.
usage: check_trace_strips.py [n [perk|fused]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import list_faithful as lf  # noqa: E402
from floydwarshall_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 516
ENGINE = {"perk": engine.FWX_ENGINE_PERK, "fused": engine.FWX_ENGINE_FUSED}[sys.argv[2] if len(sys.argv) > 2 else "perk"]
rate, nxt, hops = synth.make("t1", n, np.float64, seed=77)
vertices = [("X", "C%04d" % i) for i in range(n)]
t0 = time.time()
ref = lf.path_indices(lf.run_algo(lf.from_dense(vertices, rate, nxt), np.float64))
print("list-faithful restatement: %.0f s" % (time.time() - t0), flush=True)
with engine.DeviceMatrix(n, np.float64, with_next=True) as dm:
    dm.enable_path_log()
    dm.upload(rate, nxt)
    dm.solve(engine=ENGINE)
    _, nx, _ = dm.download()
    rnd = np.random.default_rng(1)
    differs = 0
    for _ in range(20000):
        i, j = (int(x) for x in rnd.integers(0, n, size=2))
        got = tuple(dm.query_exact(i, j)[1])
        assert got == ref[i][j], (i, j, got, ref[i][j])
        try:
            differs += list(got) != engine.follow_path(nx, i, j)
        except engine.FwxError:
            differs += 1
print("check_trace_strips: OK, n=%d, 20000 lists equal the reference's (%d differ from the next-hop walk)" % (n, differs))
