"""Lazy next-hops (fused_range in csrc/fwx_api.hip): a rates + next solve runs the rates-only double
pass, whose main kernels stamp every entry they move with the index of the pass pair, and next-hops are
resolved only where somebody needs them -- the pivot columns of a block before its column panel, and
everything once at the end -- from the panels of the stamped pair (first pivot whose product equals
the entry's value).  By default from N = 6144 on; FWX_LAZY_NEXT_MIN_N=0 forces it here at sizes the
oracle solves in seconds (orders that are multiples of 128).  Rates AND next-hops bit for bit: ties
(the earliest pivot of the last update must win), sparse inputs, overflow to +inf, f32 and f64."""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_lazy(monkeypatch):
    monkeypatch.setenv("FWX_LAZY_NEXT_MIN_N", "0")


def _check(rate, nxt):
    er, en = rate.copy(), nxt.copy()
    oracle.relax(er, en)
    gr, gn = rate.copy(), nxt.copy()
    engine.solve(gr, gn, engine=engine.FWX_ENGINE_FUSED)              # uncounted, no hops: lazy next-hops
    assert_bits_equal(gr, er, "rate")
    assert_bits_equal(gn, en, "next")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [256, 384, 512, 640, 1024, 1280])
def test_lazy_next_sizes(n, dtype):
    rate, nxt, _ = synth.make("d1", n, dtype, seed=7100 + n)
    _check(rate, nxt)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["d2", "t1", "t2", "t4"])
def test_lazy_next_distributions(kind, dtype):
    rate, nxt, _ = synth.make(kind, 768, dtype, seed=94)
    _check(rate, nxt)


def test_lazy_next_through_a_handle_twice_and_against_the_arg_kernels(monkeypatch):
    n = 896
    rate, nxt, _ = synth.make("t1", n, np.float32, seed=79)
    outs = []
    for thresh in ("0", "0", "100000000"):
        monkeypatch.setenv("FWX_LAZY_NEXT_MIN_N", thresh)
        with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
            for _ in range(2):
                dm.upload(rate, nxt)
                dm.solve()
            r, x, _ = dm.download()
            outs.append((r, x))
            src = np.arange(0, n, 3, dtype=np.int32)
            dst = ((src * 7 + 11) % n).astype(np.int32)
            for s, d in zip(src[:60], dst[:60]):       # queries walk the resolved next-hops
                q_rate, q_path = dm.query(int(s), int(d))
                assert q_rate == r[s, d] and q_path == oracle.follow_path(x, int(s), int(d))
    er, en = rate.copy(), nxt.copy()
    oracle.relax(er, en)
    for r, x in outs:
        assert_bits_equal(r, er, "rate")
        assert_bits_equal(x, en, "next")


def test_shapes_that_do_not_qualify_take_the_arg_kernels():
    """Orders that are no multiple of 128, pivot ranges, hops, the path trace, update counting: the
    lazy form steps aside and results stay the oracle's."""
    for n in (260, 452):
        rate, nxt, _ = synth.make("d2", n, np.float32, seed=n)
        _check(rate, nxt)
    rate, nxt, hops = synth.make("t1", 512, np.float32, seed=3)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    eu = oracle.relax(er, en, eh)
    gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
    engine.solve(gr, gn, gh, engine=engine.FWX_ENGINE_FUSED)
    assert_bits_equal(gr, er, "rate (hops)")
    assert_bits_equal(gn, en, "next (hops)")
    assert_bits_equal(gh, eh, "hops")
    gr, gn = rate.copy(), nxt.copy()
    assert engine.solve(gr, gn, engine=engine.FWX_ENGINE_FUSED, count_updates=True) == eu
    assert_bits_equal(gn, en, "next (counted)")
    er, en = rate.copy(), nxt.copy()
    oracle.relax(er, en, None, 128, 384)
    gr, gn = rate.copy(), nxt.copy()
    engine.solve(gr, gn, engine=engine.FWX_ENGINE_FUSED, k_begin=128, k_end=384)
    assert_bits_equal(gr, er, "rate (range)")
    assert_bits_equal(gn, en, "next (range)")
