"""The look-ahead schedules of the fused engine (fused_range in csrc/fwx_api.hip).  Small matrices
take the serial schedule (what every other GPU test exercises); from n = 4096..8192 the next
block's rows AND columns are relaxed on a side stream, beside the main kernel that skips them
(symmetric look-ahead); pivot ranges that do not start on a multiple of 64 fall back to the rows-only
look-ahead.  FWX_LOOKAHEAD_MIN_N=0 forces the look-ahead forms here at sizes the oracle solves in
seconds, for every kernel family behind them: max form (f32 / f64 rates), arg (f32 + next, + hops,
+ trace), compare form (update counting, f64 + next, inputs outside the domain)."""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth
from oracle import list_faithful as lf

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_symmetric(monkeypatch):
    monkeypatch.setenv("FWX_LOOKAHEAD_MIN_N", "0")
    monkeypatch.setenv("FWX_SYMMETRIC_MIN_N", "0")


def _check(rate, nxt, hops, count, **kw):
    er = rate.copy()
    en = None if nxt is None else nxt.copy()
    eh = None if hops is None else hops.copy()
    eu = oracle.relax(er, en, eh, kw.get("k_begin", 0), kw.get("k_end") or None)
    gr = rate.copy()
    gn = None if nxt is None else nxt.copy()
    gh = None if hops is None else hops.copy()
    u = engine.solve(gr, gn, gh, count_updates=count, engine=engine.FWX_ENGINE_FUSED, **kw)
    assert_bits_equal(gr, er, "rate")
    if nxt is not None:
        assert_bits_equal(gn, en, "next")
    if hops is not None:
        assert_bits_equal(gh, eh, "hops")
    if count:
        assert u == eu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [128, 192, 260, 452, 1000])
@pytest.mark.parametrize("count", [False, True])
def test_symmetric_schedule_all_fields(n, count, dtype):
    rate, nxt, hops = synth.make("d1", n, dtype, seed=4100 + n)
    _check(rate, None, None, count)
    _check(rate, nxt, None, count)
    _check(rate, nxt, hops, count)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["d2", "t1", "t2", "t3", "t4"])
def test_symmetric_schedule_distributions(kind, dtype):
    """Ties (the earliest pivot must win), sparse inputs, overflow, values outside the domain."""
    rate, nxt, hops = synth.make(kind, 324, dtype, seed=91)
    _check(rate, nxt, hops, False)
    _check(rate, None, None, False)
    _check(rate, nxt, None, True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_symmetric_schedule_pivot_ranges(dtype):
    """Aligned ranges take the symmetric form (ragged last block included), others fall back."""
    rate, nxt, hops = synth.make("d2", 400, dtype, seed=5)
    for kb, ke in ((0, 400), (64, 333), (128, 192), (0, 130), (37, 211)):
        _check(rate, nxt, hops, False, k_begin=kb, k_end=ke)
        _check(rate, None, None, False, k_begin=kb, k_end=ke)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_symmetric_schedule_exact_path_lists(dtype):
    """The path trace kept by rowpanel / colpanel / main under the symmetric schedule against the
    list-faithful restatement, ties and sparse inputs."""
    n = 200
    vertices = [("X", "C%03d" % i) for i in range(n)]
    for kind in ("t1", "t2"):
        rate, nxt, _ = synth.make(kind, n, dtype, seed=14)
        ref = lf.path_indices(lf.run_algo(lf.from_dense(vertices, rate, nxt), dtype))
        with engine.DeviceMatrix(n, dtype, with_next=True) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt)
            dm.solve(engine=engine.FWX_ENGINE_FUSED)
            for i in range(0, n, 3):
                for j in range(n):
                    assert tuple(dm.query_exact(i, j)[1]) == ref[i][j], (kind, i, j)


def test_symmetric_schedule_reused_handle_and_default_threshold(monkeypatch):
    """A handle solved twice (kept workspace and side stream), then the same input with the
    schedule switched off: identical bits."""
    n = 516
    rate, nxt, hops = synth.make("d1", n, np.float32, seed=77)
    outs = []
    # symmetric, symmetric again, rows-only look-ahead, serial
    for look, sym in (("0", "0"), ("0", "0"), ("0", "1000000"), ("1000000", "0")):
        monkeypatch.setenv("FWX_LOOKAHEAD_MIN_N", look)
        monkeypatch.setenv("FWX_SYMMETRIC_MIN_N", sym)
        with engine.DeviceMatrix(n, np.float32, with_next=True, with_hops=True) as dm:
            for _ in range(2):
                dm.upload(rate, nxt, hops)
                dm.solve(engine=engine.FWX_ENGINE_FUSED)
            outs.append(dm.download())
    for o in outs[1:]:
        assert_bits_equal(o[0], outs[0][0], "rate")
        assert np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("count", [False, True])
def test_rows_only_lookahead_all_fields(count, dtype, monkeypatch):
    """The fallback form (symmetric switched off), ragged sizes and an unaligned pivot range."""
    monkeypatch.setenv("FWX_SYMMETRIC_MIN_N", "1000000")
    for n in (132, 260, 516):
        rate, nxt, hops = synth.make("d2", n, dtype, seed=4300 + n)
        _check(rate, None, None, count)
        _check(rate, nxt, hops, count)
    rate, nxt, hops = synth.make("t1", 324, dtype, seed=8)
    _check(rate, nxt, hops, count, k_begin=37, k_end=300)
