#!/usr/bin/env python3
"""Long-running fuzz INSIDE the reference's domain (developer tool): matrices of non-negative rates
with exact ties, zeros (no route), subnormals, huge values whose products overflow to +inf (and then
inf * 0 = NaN candidates), consistent next-hops -- the inputs that take the max-form kernels
(fused_main_max, fused_main_arg, fused_main_max_f64) and the panels that carry hops / the path
trace.  Every run: oracle (rate, next, hops, U) against the fused engine through the host API, a
traced handle (exact `_path` lists against the per-k engine's trace), and a partitioned handle
with a random number of logical partitions.  usage: fuzz_domain.py [seconds [max_n]]"""
import os
import sys
import time

import numpy as np

if os.environ.get("FUZZ_IMPORT_TORCH") == "1":
    # torch FIRST: libfwx then binds to the HIP runtime bundled with the torch wheel (what every Python
    # caller got in rounds 1-2, and what floydwarshall_amd.dist still gets).  Default: no torch in the
    # process, libfwx on the runtime it was built against.
    import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from floydwarshall_amd import engine  # noqa: E402
from helpers import assert_bits_equal  # noqa: E402


def domain_matrix(rnd, n, dtype):
    fi = np.finfo(dtype)
    kind = int(rnd.integers(0, 5))
    if kind == 0:      # exact ties everywhere
        rate = np.ldexp(1.0, -rnd.integers(0, 4, size=(n, n))).astype(dtype)
    elif kind == 1:    # uniform, dense
        rate = (0.05 + 0.95 * rnd.random((n, n))).astype(dtype)
    elif kind == 2:    # sparse ties
        rate = np.ldexp(1.0, -rnd.integers(0, 3, size=(n, n))).astype(dtype)
        rate[rnd.random((n, n)) > 0.2] = 0
    elif kind == 3:    # overflow: products reach +inf, inf * 0 appears
        rate = np.exp(rnd.uniform(-5.0, np.log(float(fi.max)) * 0.6, size=(n, n))).astype(dtype)
        rate[rnd.random((n, n)) < 0.3] = 0
        rate[rnd.random((n, n)) < 0.01] = np.inf
    else:              # subnormal products
        rate = (rnd.random((n, n)) * float(fi.tiny) * 64).astype(dtype)
        rate[rnd.random((n, n)) < 0.5] = dtype(1.0)
    np.fill_diagonal(rate, 0)
    if rnd.random() < 0.3:                       # arbitrary (but consistent) diagonal
        d = rnd.integers(0, n, size=max(1, n // 8))
        rate[d, d] = dtype(0.5)
    nxt = np.where(rate != 0, np.arange(n, dtype=np.int32)[None, :], -1).astype(np.int32)
    hops = (nxt >= 0).astype(np.int32)
    return np.ascontiguousarray(rate), np.ascontiguousarray(nxt), np.ascontiguousarray(hops)


import faulthandler  # noqa: E402
faulthandler.enable()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 700
seed = int(sys.argv[3]) if len(sys.argv) > 3 else int(time.time())
print("fuzz_domain: seed %d" % seed, flush=True)
rnd = np.random.default_rng(seed)
trail = open(os.environ["FUZZ_TRAIL"], "w") if os.environ.get("FUZZ_TRAIL") else None


def note(msg):
    """last thing attempted, for a crash that leaves no traceback"""
    if trail is None:
        return
    trail.seek(0)
    trail.truncate()
    trail.write(msg + "\n")
    trail.flush()


t0 = time.time()
cases = 0
with np.errstate(all="ignore"):
    while time.time() - t0 < budget:
        dtype = np.float64 if rnd.random() < 0.4 else np.float32
        n = int(rnd.integers(2, max_n))
        rate, nxt, hops = domain_matrix(rnd, n, dtype)
        note("case %d: n=%d %s oracle" % (cases, n, dtype.__name__))
        er, en, eh = rate.copy(), nxt.copy(), hops.copy()
        eu = oracle.relax_mt(er, en, hops=eh, threads=8) if n > 256 else oracle.relax(er, en, eh)
        # host API, fused engine: rates only / + next / + next + hops, counted and not
        for fields in (0, 1, 2):
            gr = rate.copy()
            gn = nxt.copy() if fields >= 1 else None
            gh = hops.copy() if fields >= 2 else None
            count = bool(rnd.integers(0, 2))
            note("case %d: n=%d %s host solve fields=%d count=%s" % (cases, n, dtype.__name__, fields, count))
            u = engine.solve(gr, gn, gh, engine=engine.FWX_ENGINE_FUSED, count_updates=count)
            assert_bits_equal(gr, er, "rate n=%d %s fields=%d" % (n, dtype.__name__, fields))
            if gn is not None:
                assert_bits_equal(gn, en, "next n=%d" % n)
            if gh is not None:
                assert_bits_equal(gh, eh, "hops n=%d" % n)
            assert not count or u == eu
        # partitioned handle, random P, with hops and the trace; exact lists against the per-k trace.  Round 4:
        # the plain handle also on AUTO (any n: the handle pads its rows), the pair schedule forced at random
        # (FWX_DOUBLE_PASS_*: read on every solve), and one price change RESUMED on each resumable handle
        parts = int(rnd.integers(1, 9))
        src = rnd.integers(0, n, 64).astype(np.int32)
        dst = rnd.integers(0, n, 64).astype(np.int32)
        lists = []
        for var in ("FWX_DOUBLE_PASS_MIN_N", "FWX_DOUBLE_PASS_NEXT_MIN_N"):
            if rnd.random() < 0.5:
                os.environ[var] = "0"
            else:
                os.environ.pop(var, None)
        # the price change: one entry of a late row scaled down (stays inside the domain, no new arbitrage)
        pi, pj = int(rnd.integers(n // 2, n)), int(rnd.integers(0, n))
        patch = pi != pj and n > 64 and bool(rate[pi, pj] > 0) and bool(np.isfinite(rate[pi, pj]))
        if patch:
            r2 = rate.copy()
            r2[pi, pj] = dtype(r2[pi, pj] * dtype(0.75))
            er2, en2, eh2 = r2.copy(), nxt.copy(), hops.copy()
            note("case %d: n=%d %s oracle of the patched input" % (cases, n, dtype.__name__))
            if n > 256:
                oracle.relax_mt(er2, en2, hops=eh2, threads=8)
            else:
                oracle.relax(er2, en2, eh2)
        for kw in (dict(devices=[0] * parts), dict(device=0), dict(device=0, auto=True)):
            note("case %d: n=%d %s handle %s" % (cases, n, dtype.__name__, kw))
            auto = kw.pop("auto", False)
            with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True, **kw) as dm:
                dm.enable_path_log()
                resumable = patch and ("devices" in kw or auto)
                if resumable:
                    dm.keep_input()
                    dm.enable_resume(3)
                dm.upload(rate, nxt, hops)
                dm.solve(engine=engine.FWX_ENGINE_AUTO if ("devices" in kw or auto) else engine.FWX_ENGINE_PERK)
                gr, gn, gh = dm.download()
                assert_bits_equal(gr, er, "handle rate n=%d P=%s" % (n, kw))
                assert_bits_equal(gn, en, "handle next")
                assert_bits_equal(gh, eh, "handle hops")
                lists.append(dm.query_exact_batch(src, dst, cap=16 * n + 64))
                if resumable:
                    note("case %d: n=%d %s resolve (%d,%d) %s" % (cases, n, dtype.__name__, pi, pj, kw))
                    dm.resolve(np.array([pi * n + pj], dtype=np.int64), np.array([r2[pi, pj]], dtype=dtype),
                               np.array([pj], dtype=np.int32), np.array([1], dtype=np.int32))
                    gr, gn, gh = dm.download()
                    assert_bits_equal(gr, er2, "resumed rate n=%d %s" % (n, kw))
                    assert_bits_equal(gn, en2, "resumed next")
                    assert_bits_equal(gh, eh2, "resumed hops")
        assert lists[0] == lists[1] == lists[2], "exact lists: partitioned / plain fused trace vs per-k trace, n=%d P=%d" % (n, parts)
        cases += 1
        if cases % 20 == 0:
            print("%d cases, %.0f s" % (cases, time.time() - t0), flush=True)
print("fuzz_domain: OK, %d cases in %.0f s" % (cases, time.time() - t0))
