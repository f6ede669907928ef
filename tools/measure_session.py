#!/usr/bin/env python3
"""Latency of the host mirror (C++ Session over the GPU engine): rate update -> first best-rate
query (buildMatrix + traced solve + exact path), then cached queries."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from floydwarshall_amd import host  # noqa: E402


def market(n_exch, n_ccy, seed=1):
    rnd = np.random.default_rng(seed)
    ccys = ["C%02d" % i for i in range(n_ccy)]
    price = dict(zip(ccys, 0.5 + 1.5 * rnd.random(n_ccy)))
    rows = []
    for e in range(n_exch):
        exch = "X" + "".join(chr(65 + (e // 26 ** p) % 26) for p in range(3))
        for i in range(n_ccy):
            for j in range(i + 1, n_ccy):
                if rnd.random() < 0.6:
                    a, b = ccys[i], ccys[j]
                    rows.append((1000 + e, exch, a, b, price[b] / price[a] * (0.97 + 0.03 * rnd.random()),
                                 price[a] / price[b] * (0.97 + 0.03 * rnd.random())))
    return rows


for n_exch, n_ccy in ((2, 2), (6, 8), (20, 12), (60, 16), (128, 16)):
    rows = market(n_exch, n_ccy)
    s = host.Session(device=0)
    for r in rows:
        s.update_rates(*r)
    vertices, _, _ = s.build_matrix()
    n = len(vertices)
    a, b = vertices[0], vertices[-1]
    t0 = time.perf_counter()
    try:
        s.find_best_rate(a, b)
    except host.AlgoError:
        pass
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    reps = 50
    for q in range(reps):
        try:
            s.find_best_rate(vertices[q % n], vertices[(q * 7 + 3) % n])
        except host.AlgoError:
            pass
    t_cached = (time.perf_counter() - t0) / reps
    # one more rate change -> re-solve
    s.update_rates(99999, rows[0][1], rows[0][2], rows[0][3], rows[0][4] * 0.999, rows[0][5])
    t0 = time.perf_counter()
    try:
        s.find_best_rate(a, b)
    except host.AlgoError:
        pass
    t_resolve = time.perf_counter() - t0
    print("n=%5d vertices (%d rates): first query %.2f ms, re-solve after a rate change %.2f ms, "
          "cached query %.3f ms" % (n, s.rate_count, 1e3 * t_first, 1e3 * t_resolve, 1e3 * t_cached),
          flush=True)
