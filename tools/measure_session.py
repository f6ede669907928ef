#!/usr/bin/env python3
"""Latency of the host mirror (C++ Session over the GPU engine): rate update -> first best-rate
query (buildMatrix + traced solve + exact path), cached queries, and the re-solve after a price
change between known vertices -- from pivot 0 (checkpoints off) and RESUMED at the last checkpoint
the changed entries cannot have influenced (fwx_matrix_resolve), by position of the changed pair."""
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
        for i in range(n_ccy):                     # a ring (every currency appears: n = n_exch * n_ccy) ...
            j = (i + 1) % n_ccy
            a, b = ccys[min(i, j)], ccys[max(i, j)]
            rows.append((1000 + e, exch, a, b, price[b] / price[a] * (0.97 + 0.03 * rnd.random()),
                         price[a] / price[b] * (0.97 + 0.03 * rnd.random())))
            for j in range(i + 2, n_ccy):          # ... plus random chords
                if rnd.random() < 0.5 and (i, j) != (0, n_ccy - 1):
                    a, b = ccys[i], ccys[j]
                    rows.append((1000 + e, exch, a, b, price[b] / price[a] * (0.97 + 0.03 * rnd.random()),
                                 price[a] / price[b] * (0.97 + 0.03 * rnd.random())))
    return rows


def timed_query(s, a, b):
    t0 = time.perf_counter()
    try:
        s.find_best_rate(a, b)
    except host.AlgoError:
        pass
    return time.perf_counter() - t0


# (27 x 9 = 243 and 113 x 9 = 1017 vertices: ODD orders -- the handle pads its rows on the device: the fused
# engine instead of one launch per pivot, and resumable like the even orders)
# usage: measure_session.py [--devices 0,0,0,0]   (the resident matrix row-partitioned: it resumes too)
DEVICES = None
if "--devices" in sys.argv:
    DEVICES = [int(x) for x in sys.argv[sys.argv.index("--devices") + 1].split(",")]
for n_exch, n_ccy in ((2, 2), (6, 8), (20, 12), (27, 9), (60, 16), (113, 9), (128, 16), (256, 16)):
    rows = market(n_exch, n_ccy)
    line = {}
    for label, cps in (("full", 0), ("resumable", 7)):
        s = host.Session(device=0)
        if DEVICES:
            s.set_devices(DEVICES, min_vertices=65)
        s.set_checkpoints(cps)
        for r in rows:
            s.update_rates(*r)
        vertices, _, _ = s.build_matrix()
        n = len(vertices)
        a, b = vertices[0], vertices[-1]
        t_first = timed_query(s, a, b)
        reps = 50
        t0 = time.perf_counter()
        for q in range(reps):
            timed_query(s, vertices[q % n], vertices[(q * 7 + 3) % n])
        t_cached = (time.perf_counter() - t0) / reps
        # price changes on the exchange that owns the first, the middle and the last vertex of the matrix
        by_pos = {}
        for pos, vi in (("first", 0), ("middle", n // 2), ("last", n - 1)):
            old = [r for r in rows if r[1] == vertices[vi][0]][0]     # vertices are in matrix order
            ts = []
            for rep in range(5):
                s.update_rates(200000 + rep, old[1], old[2], old[3], old[4] * (0.999 - 1e-4 * rep), old[5])
                ts.append(timed_query(s, a, b))
            by_pos[pos] = 1e3 * sorted(ts)[len(ts) // 2]
        line[label] = (n, s.rate_count, t_first, t_cached, by_pos, s.resumed_solves, s.resumed_pivots, s.patched_solves)
    n, nr, t_first, t_cached, _, _, _, _ = line["full"]
    f, r = line["full"][4], line["resumable"][4]
    print("n=%5d vertices (%d rates): first query %.2f ms (resumable handle %.2f), cached query %.3f ms; re-solve "
          "after a price change on the first / middle / last exchange: from pivot 0 %.2f / %.2f / %.2f ms, "
          "resumed %.2f / %.2f / %.2f ms (%d of %d patched solves resumed, %d pivots skipped)"
          % (n, nr, 1e3 * t_first, 1e3 * line["resumable"][2], 1e3 * t_cached, f["first"], f["middle"], f["last"],
             r["first"], r["middle"], r["last"], line["resumable"][5], line["resumable"][7], line["resumable"][6]),
          flush=True)
