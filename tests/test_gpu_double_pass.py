"""The double-pass schedule (fused_range in csrc/fwx_api.hip): the max-form main kernels (rates only)
and the arg kernels (rates + next-hops, + path trace, + hops) apply TWO passes = 128 pivots per launch,
the side stream keeps two passes' worth of panels ahead.  By default from N = 6144 (rates only) and
N = 5120 (with next-hops) on; FWX_DOUBLE_PASS_MIN_N=0 / FWX_DOUBLE_PASS_NEXT_MIN_N=0 force it here at
sizes the oracle solves in seconds: even / odd numbers of 64-blocks, ragged tails, matrix orders that
are no multiple of the tile, aligned pivot ranges, ties / sparse / overflowing inputs, f32 and f64.
Bit for bit, exact `_path` lists included."""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_double_pass(monkeypatch):
    monkeypatch.setenv("FWX_DOUBLE_PASS_MIN_N", "0")
    monkeypatch.setenv("FWX_DOUBLE_PASS_NEXT_MIN_N", "0")


def _check(rate, **kw):
    er = rate.copy()
    oracle.relax(er, None, None, kw.get("k_begin", 0), kw.get("k_end") or None)
    gr = rate.copy()
    engine.solve(gr, engine=engine.FWX_ENGINE_FUSED, **kw)           # rates only, uncounted: max form
    assert_bits_equal(gr, er, "rate")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [256, 260, 320, 384, 452, 512, 708, 1000, 1284])
def test_double_pass_sizes(n, dtype):
    rate, _, _ = synth.make("d1", n, dtype, seed=6100 + n)
    _check(rate)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["d2", "t1", "t2", "t4"])
def test_double_pass_distributions(kind, dtype):
    rate, _, _ = synth.make(kind, 644, dtype, seed=93)
    _check(rate)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_double_pass_pivot_ranges(dtype):
    """Aligned ranges of at least four blocks take the double pass (odd block counts and ragged ends
    included); shorter or unaligned ones the other schedules."""
    rate, _, _ = synth.make("d2", 900, dtype, seed=7)
    for kb, ke in ((0, 900), (64, 333), (128, 900), (0, 256), (0, 320), (192, 517), (37, 611), (0, 200)):
        _check(rate, k_begin=kb, k_end=ke)


def test_double_pass_through_a_handle_twice_and_against_the_single_pass(monkeypatch):
    n = 772
    rate, _, _ = synth.make("d1", n, np.float32, seed=78)
    outs = []
    for thresh in ("0", "0", "100000000"):
        monkeypatch.setenv("FWX_DOUBLE_PASS_MIN_N", thresh)
        with engine.DeviceMatrix(n, np.float32, with_next=False) as dm:
            for _ in range(2):
                dm.upload(rate)
                dm.solve()
            outs.append(dm.download()[0])
    er = rate.copy()
    oracle.relax(er)
    for o in outs:
        assert_bits_equal(o, er, "rate")


# ---- with next-hops: the arg kernels run their two passes on the tile they keep in registers ----------
def _check_next(rate, nxt, hops=None, **kw):
    er, en = rate.copy(), nxt.copy()
    eh = None if hops is None else hops.copy()
    oracle.relax(er, en, eh, kw.get("k_begin", 0), kw.get("k_end") or None)
    gr, gn = rate.copy(), nxt.copy()
    gh = None if hops is None else hops.copy()
    engine.solve(gr, gn, gh, engine=engine.FWX_ENGINE_FUSED, **kw)   # uncounted, inside the domain: arg form
    assert_bits_equal(gr, er, "rate")
    assert_bits_equal(gn, en, "next")
    if hops is not None:
        assert_bits_equal(gh, eh, "hops")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [256, 260, 320, 384, 452, 512, 708, 1000, 1284])
def test_double_pass_with_next_hops_sizes(n, dtype):
    rate, nxt, hops = synth.make("d1", n, dtype, seed=6200 + n)
    _check_next(rate, nxt)
    if n in (260, 512, 708):
        _check_next(rate, nxt, hops)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["t1", "t2", "t4"])
def test_double_pass_with_next_hops_distributions(kind, dtype):
    """Ties (the first pivot of a pass that attains the maximum wins: the re-scan of BOTH passes of a
    launch must see its own pass's strips), sparse and overflowing inputs; hops ride along."""
    rate, nxt, hops = synth.make(kind, 644, dtype, seed=94)
    _check_next(rate, nxt, hops)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_double_pass_with_next_hops_pivot_ranges(dtype):
    rate, nxt, _ = synth.make("d1", 900, dtype, seed=8)
    for kb, ke in ((0, 900), (64, 333), (128, 900), (0, 256), (0, 320), (192, 517), (37, 611)):
        _check_next(rate, nxt, k_begin=kb, k_end=ke)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_double_pass_with_the_path_trace_gives_the_reference_lists(dtype, monkeypatch):
    """The traced double-pass solve: rates and next-hops equal the oracle's, the exact `_path` lists
    (Algorithms.hs:55) rebuilt from last / at_col / at_row equal the list-faithful restatement's -- on a
    tie-heavy input, where they differ from next-hop walks -- and the single-pass schedule's."""
    from oracle import list_faithful as lf
    # (the list-faithful restatement is pure Python, and slow on float32 scalars: four blocks + a ragged
    #  tail for f32, five -- an odd last block -- for f64)
    n = 260 if dtype == np.float32 else 324
    rate, nxt, _ = synth.make("t1", n, dtype, seed=11)
    m = lf.run_algo(lf.from_dense([("X", "C%03d" % i) for i in range(n)], rate, nxt), dtype)
    paths = lf.path_indices(m)
    er, en = rate.copy(), nxt.copy()
    oracle.relax(er, en)
    rnd = np.random.default_rng(12)
    src = rnd.integers(0, n, 600).astype(np.int32)
    dst = rnd.integers(0, n, 600).astype(np.int32)
    for thresh in ("0", "100000000"):
        monkeypatch.setenv("FWX_DOUBLE_PASS_NEXT_MIN_N", thresh)
        with engine.DeviceMatrix(n, dtype, with_next=True) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt)
            dm.solve(engine=engine.FWX_ENGINE_FUSED)
            gr, gn, _ = dm.download()
            got = dm.query_exact_batch(src, dst)
        assert_bits_equal(gr, er, "rate")
        assert_bits_equal(gn, en, "next")
        for q in range(len(src)):
            assert tuple(got[q]) == paths[src[q]][dst[q]]
