#!/usr/bin/env python3
"""Long-running fuzz (developer tool, not part of the test-suite): hostile-valued matrices through
every engine against the dense oracle, and traced solves (exact `_path` lists) against the list-faithful restatement.
usage: fuzz_long.py [seconds [max_n]]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from oracle import list_faithful as lf  # noqa: E402
from floydwarshall_amd import engine  # noqa: E402
from helpers import assert_bits_equal  # noqa: E402
from test_gpu_parity import _hostile_matrix, _solve_and_compare  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rnd = np.random.default_rng(int(time.time()))
t0 = time.time()
cases = logged = 0
with np.errstate(all="ignore"):
    while time.time() - t0 < budget:
        dtype = np.float64 if rnd.random() < 0.5 else np.float32
        n = int(rnd.integers(1, max_n))
        rate, nxt, hops = _hostile_matrix(rnd, n, dtype)
        _solve_and_compare(rate, nxt, hops)
        _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_PERK)
        _solve_and_compare(rate, nxt, None, engine=engine.FWX_ENGINE_FUSED)
        _solve_and_compare(rate, None, None, engine=engine.FWX_ENGINE_FUSED)
        cases += 1
        if cases % 5 == 0:
            # traced solve vs the list-faithful `_path` lists (small n: the python restatement is O(n^3))
            m = 2 * int(rnd.integers(1, 20)) if cases % 25 else 2 * int(rnd.integers(33, 80))
            rate, nxt, hops = _hostile_matrix(rnd, m, np.float64)
            nxt[np.arange(m), np.arange(m)] = -1
            hops = (nxt >= 0).astype(np.int32)
            vertices = [("X", "C%03d" % i) for i in range(m)]
            ref = lf.run_algo(lf.from_dense(vertices, rate, nxt), np.float64)
            ref_paths = lf.path_indices(ref)
            er, en, eh = rate.copy(), nxt.copy(), hops.copy()
            oracle.relax(er, en, eh)
            for eng in (engine.FWX_ENGINE_AUTO, engine.FWX_ENGINE_PERK, engine.FWX_ENGINE_FUSED):
                with engine.DeviceMatrix(m, np.float64, with_next=True, with_hops=True) as dm:
                    dm.enable_path_log()
                    dm.upload(rate, nxt, hops)
                    dm.solve(engine=eng)
                    dm.upload(rate, nxt, hops)       # second solve on the same handle
                    dm.solve(engine=eng)
                    gr, gn, hp = dm.download()       # FUSED: hops carried through the panels
                    assert_bits_equal(gr, er, "rate")
                    assert np.array_equal(gn, en) and np.array_equal(hp, eh)
                    for i in range(m):
                        for j in range(m):
                            if hp[i, j] > 8 * m:
                                continue
                            assert tuple(dm.query_exact(i, j, cap=16 * m + 64)[1]) == ref_paths[i][j], (m, i, j)
            logged += 1
        if cases % 200 == 0:
            print("%d cases, %d logged cases, %.0f s" % (cases, logged, time.time() - t0), flush=True)
print("fuzz_long: OK, %d cases (%d logged) in %.0f s" % (cases, logged, time.time() - t0))
