"""The double pass on PARTITIONED handles (multi_double_pass in csrc/fwx_multi.hip): every partition's main
kernel applies two passes = 128 pivots per launch, the side streams keep two blocks' worth of panels
ahead across owners (four panel sets per partition, one exchange per block).  By default from N = 6144
(rates only) / N = 8192 (f32 with next-hops) on partitions whose first row is a multiple of 64;
FWX_DOUBLE_PASS_MIN_N=0 / FWX_DOUBLE_PASS_NEXT_MIN_N=0 force it here at sizes the oracle solves in
seconds.  Logical partitions of the one GPU there is (both transports that can run on it).  Everything bit
for bit against the oracle; exact `_path` lists against the single-device trace."""
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


def _solve_and_check(rate, nxt, hops, parts, took_pairs=True, **kw):
    er = rate.copy()
    en = None if nxt is None else nxt.copy()
    eh = None if hops is None else hops.copy()
    oracle.relax(er, en, eh, kw.get("k_begin", 0), kw.get("k_end") or None)
    n = rate.shape[0]
    with engine.DeviceMatrix(n, rate.dtype, with_next=nxt is not None, with_hops=hops is not None,
                             devices=[0] * parts, exchange=kw.pop("exchange", engine.FWX_XCHG_PEER)) as dm:
        dm.set_timing(True)
        dm.upload(rate, nxt, hops)
        dm.solve(**kw)
        t = dm.timing()
        gr, gn, gh = dm.download()
    assert_bits_equal(gr, er, "rate P=%d" % parts)
    if nxt is not None:
        assert_bits_equal(gn, en, "next P=%d" % parts)
    if hops is not None:
        assert_bits_equal(gh, eh, "hops P=%d" % parts)
    # the schedule really was the one under test
    assert t["pivots_per_step"] == (128 if took_pairs else 64), t
    assert t["steps"] >= 1 and t["bulk_us"] > 0 and t["chain_us"] > 0 and t["partitions"] == parts, t
    return t


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("parts,n", [(1, 708), (2, 640), (3, 576), (4, 1024), (8, 512), (8, 1536)])
def test_partitioned_double_pass_rates_and_next_hops(parts, n, dtype):
    """Even and odd block counts (640 / 64 = 10, 576 / 64 = 9), pairs that straddle two owners (P = 3:
    three blocks per partition; P = 8, n = 512: one block each), a ragged tail (P = 1, n = 708)."""
    rate, nxt, hops = synth.make("d1", n, dtype, seed=6400 + n + parts)
    _solve_and_check(rate, None, None, parts)
    _solve_and_check(rate, nxt, None, parts)
    if n in (640, 512):
        _solve_and_check(rate, nxt, hops, parts)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["d2", "t1", "t2", "t4"])
def test_partitioned_double_pass_distributions(kind, dtype):
    """Ties (the earliest pivot must win across the two passes of a launch), sparse and overflowing
    inputs; hops ride along."""
    rate, nxt, hops = synth.make(kind, 768, dtype, seed=95)
    _solve_and_check(rate, None, None, 4)
    _solve_and_check(rate, nxt, hops, 4)
    _solve_and_check(rate, nxt, None, 2, exchange=engine.FWX_XCHG_PEER)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_partitioned_double_pass_pivot_ranges_and_fallbacks(dtype):
    """Aligned ranges of at least four full blocks take the pairs -- a ragged end then goes through the
    single-pass loop in the same solve --, anything else the single pass from the start: ranges that do
    not start on a block, fewer than four blocks, a counted solve (U: the compare form).  Partitions are cut
    on multiples of 64 (n >= 128 P), so the order of the matrix does not matter."""
    rate, nxt, _ = synth.make("d2", 1024, dtype, seed=9)
    for kb, ke, pairs in ((0, 1024, True), (0, 500, True), (128, 900, True), (64, 333, True), (512, 1024, True),
                          (37, 611, False), (0, 200, False), (960, 1024, False)):
        _solve_and_check(rate, nxt, None, 2, took_pairs=pairs, k_begin=kb, k_end=ke)
        _solve_and_check(rate, None, None, 4, took_pairs=pairs, k_begin=kb, k_end=ke)
    odd, onx, _ = synth.make("d1", 1156, dtype, seed=10)             # 1156 / 2 = 578: the library cuts at 576
    _solve_and_check(odd, onx, None, 2, took_pairs=True)             # (64-aligned partitions for n >= 128 P: any n)
    _solve_and_check(odd, None, None, 3, took_pairs=True)            # 0, 384, 768
    _solve_and_check(odd, None, None, 1, took_pairs=True)
    er, en = rate.copy(), nxt.copy()
    eu = oracle.relax(er, en)
    gr, gn = rate.copy(), nxt.copy()
    u = engine.solve_multi(gr, gn, devices=[0] * 4, count_updates=True)
    assert u == eu
    assert_bits_equal(gr, er, "counted rate")
    assert_bits_equal(gn, en, "counted next")


def test_partitioned_double_pass_over_rccl_with_one_device():
    """The RCCL transport (grouped ncclBroadcast per panel on the side streams) under the pair schedule --
    with the one-device communicator this box allows."""
    rate, nxt, _ = synth.make("d1", 640, np.float32, seed=21)
    _solve_and_check(rate, nxt, None, 1, exchange=engine.FWX_XCHG_RCCL)
    _solve_and_check(rate, None, None, 1, exchange=engine.FWX_XCHG_RCCL)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_partitioned_double_pass_keeps_the_path_trace(dtype):
    """The traced pair schedule on slabs: rates and next-hops equal the oracle's, and the exact `_path`
    lists (Algorithms.hs:55) rebuilt from the slab-local last / at_col / at_row equal those of the
    single-device traced solve (itself tied to the list-faithful restatement in test_gpu_double_pass.py)
    -- on a tie-heavy input, where they differ from next-hop walks."""
    n = 512
    rate, nxt, hops = synth.make("t1", n, dtype, seed=13)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    oracle.relax(er, en, eh)
    rnd = np.random.default_rng(14)
    src = rnd.integers(0, n, 800).astype(np.int32)
    dst = rnd.integers(0, n, 800).astype(np.int32)
    lists = []
    for devices in (None, [0] * 2, [0] * 8):
        with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True, devices=devices) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt, hops)
            dm.solve()
            gr, gn, gh = dm.download()
            lists.append(dm.query_exact_batch(src, dst))
        assert_bits_equal(gr, er, "rate")
        assert_bits_equal(gn, en, "next")
        assert_bits_equal(gh, eh, "hops")
    assert lists[1] == lists[0] and lists[2] == lists[0]
    for q in range(len(src)):
        assert len(lists[0][q]) == eh[src[q], dst[q]]


def test_timing_of_the_single_pass_and_of_the_per_k_engine(monkeypatch):
    """fwx_matrix_set_timing on the other schedules: 64 pivots per step, a chain made of look-ahead rows +
    panel kernel + exchange; single-device handles say FWX_ERR_UNSUPPORTED."""
    from floydwarshall_amd._lib import FWX_ERR_UNSUPPORTED
    monkeypatch.setenv("FWX_DOUBLE_PASS_MIN_N", "100000000")
    monkeypatch.setenv("FWX_DOUBLE_PASS_NEXT_MIN_N", "100000000")
    rate, nxt, _ = synth.make("d1", 512, np.float32, seed=5)
    t = _solve_and_check(rate, nxt, None, 4, took_pairs=False)
    assert t["steps"] == 8 and t["lookahead_us"] > 0 and t["panel_us"] > 0 and t["exchange_us"] > 0
    assert abs(t["chain_us"] - (t["lookahead_us"] + t["panel_us"] + t["exchange_us"])) < 1e-3 * t["chain_us"] + 1e-3
    t = _solve_and_check(rate, nxt, None, 4, took_pairs=False, engine=engine.FWX_ENGINE_PERK)
    assert t["steps"] == 8 and t["bulk_us"] > t["panel_us"] > 0
    with engine.DeviceMatrix(512, np.float32, with_next=False) as dm:
        with pytest.raises(engine.FwxError) as e:
            dm.set_timing(True)
        assert e.value.status == FWX_ERR_UNSUPPORTED
