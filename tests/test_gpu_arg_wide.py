"""fused_main_arg_wide: the 128 x 128-tile form of the arg re-scan kernel (f32 rates + next-hops,
+ hops, + path trace; packed stage tracking).  By default it only runs from ~N = 12288 on;
FWX_ARG_WIDE_MIN_TILES=0 forces it here at sizes the oracle solves in seconds -- ragged edges, short
passes (pivot ranges that end inside a 16-pivot stage), ties, sparse and overflowing inputs, the
look-ahead schedules (row / column skips), row-partitioned slabs.  Everything bit for bit."""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth
from oracle import list_faithful as lf

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_wide(monkeypatch):
    monkeypatch.setenv("FWX_ARG_WIDE_MIN_TILES", "0")


def _check(rate, nxt, hops, **kw):
    er, en = rate.copy(), nxt.copy()
    eh = None if hops is None else hops.copy()
    oracle.relax(er, en, eh, kw.get("k_begin", 0), kw.get("k_end") or None)
    gr, gn = rate.copy(), nxt.copy()
    gh = None if hops is None else hops.copy()
    engine.solve(gr, gn, gh, engine=engine.FWX_ENGINE_FUSED, **kw)     # uncounted: the arg kernels
    assert_bits_equal(gr, er, "rate")
    assert_bits_equal(gn, en, "next")
    if hops is not None:
        assert_bits_equal(gh, eh, "hops")


@pytest.mark.parametrize("n", [128, 132, 256, 260, 452, 1000, 1284])
def test_wide_tiles_all_fields(n):
    rate, nxt, hops = synth.make("d1", n, np.float32, seed=5100 + n)
    _check(rate, nxt, None)
    _check(rate, nxt, hops)


@pytest.mark.parametrize("kind", ["d2", "t1", "t2", "t4"])
def test_wide_tiles_distributions(kind):
    """Ties (the earliest pivot of the newest stage must win), sparse inputs, overflow to +inf."""
    rate, nxt, hops = synth.make(kind, 580, np.float32, seed=92)
    _check(rate, nxt, hops)


def test_wide_tiles_short_passes_and_pivot_ranges():
    """bt < 64: fewer than four tracking stages, or a last stage that ends early."""
    rate, nxt, hops = synth.make("t1", 400, np.float32, seed=6)
    for kb, ke in ((0, 400), (64, 333), (128, 137), (0, 17), (192, 241), (37, 211)):
        _check(rate, nxt, hops, k_begin=kb, k_end=ke)


@pytest.mark.parametrize("look,sym", [("0", "0"), ("0", "1000000")])
def test_wide_tiles_under_the_lookahead_schedules(look, sym, monkeypatch):
    monkeypatch.setenv("FWX_LOOKAHEAD_MIN_N", look)
    monkeypatch.setenv("FWX_SYMMETRIC_MIN_N", sym)
    for n in (260, 516, 1000):
        rate, nxt, hops = synth.make("d2", n, np.float32, seed=4400 + n)
        _check(rate, nxt, hops)
    rate, nxt, hops = synth.make("t1", 324, np.float32, seed=8)
    _check(rate, nxt, hops, k_begin=37, k_end=300)


def test_wide_tiles_exact_path_lists():
    n = 264
    vertices = [("X", "C%03d" % i) for i in range(n)]
    for kind in ("t1", "t2"):
        rate, nxt, _ = synth.make(kind, n, np.float32, seed=15)
        ref = lf.path_indices(lf.run_algo(lf.from_dense(vertices, rate, nxt), np.float32))
        with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt)
            dm.solve(engine=engine.FWX_ENGINE_FUSED)
            src = np.repeat(np.arange(0, n, 5, dtype=np.int32), n)
            dst = np.tile(np.arange(n, dtype=np.int32), len(range(0, n, 5)))
            got = dm.query_exact_batch(src, dst)
            for q in range(len(src)):
                assert tuple(got[q]) == ref[src[q]][dst[q]], (kind, src[q], dst[q])


@pytest.mark.parametrize("parts", [2, 3])
def test_wide_tiles_on_row_partitions(parts):
    rate, nxt, hops = synth.make("d1", 900, np.float32, seed=33)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    oracle.relax(er, en, eh)
    gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
    engine.solve_multi(gr, gn, gh, devices=[0] * parts)
    assert_bits_equal(gr, er, "rate")
    assert_bits_equal(gn, en, "next")
    assert_bits_equal(gh, eh, "hops")
