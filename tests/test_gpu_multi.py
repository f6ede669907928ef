"""The row-partitioned solve behind the C ABI (VERDICT r1 row e'): ONE process, P partitions,
fwx_matrix_create_multi / fwx_solve_multi_*.  Only one GPU is reachable here, so the partitions are
LOGICAL (the same device listed P times, panel exchange by device-to-device copy) -- the schedule,
the look-ahead, the slab kernels and the event ordering are the ones P real devices run; the RCCL
transport is exercised with a 1-device communicator (ncclCommInitAll + ncclBroadcast, root = self).
Everything is compared with the oracle bit for bit.
"""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth
from floydwarshall_amd._lib import FWX_ERR_INVALID, FWX_ERR_UNSUPPORTED
from oracle import list_faithful as lf

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def _expect(rate, nxt):
    er = rate.copy()
    en = None if nxt is None else nxt.copy()
    eu = oracle.relax(er, en)
    return er, en, eu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_logical_partitions_equal_the_oracle(parts, dtype):
    """P logical partitions through the C entry point == oracle (rates + next, U), several kinds
    and sizes incl. sizes that are not a multiple of 64, of P, or of the vector width."""
    for kind, n in (("d1", 512), ("t1", 300), ("t2", 257), ("d2", 1000)):
        rate, nxt, _ = synth.make(kind, n, dtype, seed=parts * 100 + n)
        er, en, eu = _expect(rate, nxt)
        gr, gn = rate.copy(), nxt.copy()
        u = engine.solve_multi(gr, gn, devices=[0] * parts, exchange=engine.FWX_XCHG_PEER, count_updates=True)
        assert_bits_equal(gr, er, "rate %s n=%d P=%d" % (kind, n, parts))
        assert_bits_equal(gn, en, "next %s n=%d P=%d" % (kind, n, parts))
        assert u == eu
        gr = rate.copy()                                # rates only: the max-form kernels on slabs
        engine.solve_multi(gr, devices=[0] * parts)
        assert_bits_equal(gr, er, "rates-only %s n=%d P=%d" % (kind, n, parts))


def test_more_partitions_than_panels_and_tiny_matrices():
    for n, parts in ((5, 8), (64, 3), (130, 7), (1, 2)):
        rate, nxt, _ = synth.make("d2", n, np.float64, seed=n)
        er, en, _ = _expect(rate, nxt)
        gr, gn = rate.copy(), nxt.copy()
        engine.solve_multi(gr, gn, devices=[0] * parts)
        assert_bits_equal(gr, er, "rate n=%d P=%d" % (n, parts))
        assert_bits_equal(gn, en, "next n=%d P=%d" % (n, parts))
    engine.solve_multi(np.zeros((0, 0)), devices=[0, 0])


def test_rccl_transport_with_a_one_device_communicator():
    """FWX_XCHG_RCCL: librccl.so.1 is loaded on first use, ncclCommInitAll over the device list and
    one grouped ncclBroadcast per panel on the side stream -- here with the one device there is."""
    n = 640
    rate, nxt, _ = synth.make("d1", n, np.float32, seed=77)
    er, en, eu = _expect(rate, nxt)
    with engine.DeviceMatrix(n, np.float32, with_next=True, devices=[0], exchange=engine.FWX_XCHG_RCCL) as dm:
        assert dm.parts() == (1, engine.FWX_XCHG_RCCL)
        dm.upload(rate, nxt)
        u = dm.solve(count_updates=True)
        gr, gn, _ = dm.download()
    assert_bits_equal(gr, er, "rate")
    assert_bits_equal(gn, en, "next")
    assert u == eu
    with pytest.raises(engine.FwxError) as e:            # RCCL cannot put two ranks on one device
        engine.DeviceMatrix(n, np.float32, devices=[0, 0], exchange=engine.FWX_XCHG_RCCL)
    assert e.value.status == FWX_ERR_INVALID
    with engine.DeviceMatrix(n, np.float32, devices=[0, 0]) as dm:   # AUTO falls back to peer copies
        assert dm.parts() == (2, engine.FWX_XCHG_PEER)


def test_partitioned_handle_queries_and_exact_paths():
    """A multi handle is an fwx_matrix: query walks next-hops across slabs, and with the path trace
    (kept slab-local by the same kernels) query_exact rebuilds the reference's `_path` lists."""
    n = 200
    rate, nxt, _ = synth.make("t1", n, np.float64, seed=3)     # ties: lists differ from next-hop walks
    vertices = [("X", "C%03d" % i) for i in range(n)]
    m = lf.run_algo(lf.from_dense(vertices, rate, nxt), np.float64)
    paths = lf.path_indices(m)
    er, en, _ = _expect(rate, nxt)
    with engine.DeviceMatrix(n, np.float64, with_next=True, devices=[0, 0, 0]) as dm:
        dm.enable_path_log()
        dm.upload(rate, nxt)
        with pytest.raises(engine.FwxError):
            dm.query_exact(0, 1)                                   # no traced solve yet
        dm.solve()
        gr, gn, _ = dm.download()
        assert_bits_equal(gr, er, "rate")
        assert_bits_equal(gn, en, "next")
        rnd = np.random.default_rng(5)
        src = rnd.integers(0, n, 300).astype(np.int32)
        dst = rnd.integers(0, n, 300).astype(np.int32)
        got = dm.query_exact_batch(src, dst)
        for q in range(len(src)):
            assert tuple(got[q]) == paths[src[q]][dst[q]]
        for s, d in ((0, 199), (150, 3), (70, 71)):
            r, p = dm.query_exact(s, d)
            assert r == er[s, d] and tuple(p) == paths[s][d]
            r2, p2 = dm.query(s, d)
            assert r2 == er[s, d] and p2 == oracle.follow_path(en, s, d)
        with pytest.raises(engine.FwxError):                       # a traced solve needs a fresh upload
            dm.solve()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hops_travel_with_the_panels(dtype):
    """rate + next + hops through P logical partitions (config 5's fields): hops' = hops[i][k] +
    hops[k][j] needs the hops of pivot row k at time k, which the owner exports beside the rate
    snapshot and which is exchanged with it.  f32 takes the max-form + arg re-scan kernel, f64 and
    counted solves the compare form."""
    for kind, n, parts in (("t1", 384, 3), ("d2", 1000, 2), ("t2", 200, 8)):
        rate, nxt, hops = synth.make(kind, n, dtype, seed=n)
        er, en, eh = rate.copy(), nxt.copy(), hops.copy()
        eu = oracle.relax(er, en, eh)
        for count in (False, True):
            gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
            u = engine.solve_multi(gr, gn, gh, devices=[0] * parts, count_updates=count)
            assert_bits_equal(gr, er, "rate %s" % kind)
            assert_bits_equal(gn, en, "next %s" % kind)
            assert_bits_equal(gh, eh, "hops %s" % kind)
            assert not count or u == eu


def test_unsupported_combinations_say_so():
    n = 128
    rate, nxt, hops = synth.make("d1", n, np.float32, seed=1)
    bad = rate.copy()
    bad[3, 5] = -1.0                                               # outside the reference's domain
    with pytest.raises(engine.FwxError) as e:
        engine.solve_multi(bad, nxt.copy(), devices=[0, 0])
    assert e.value.status == FWX_ERR_UNSUPPORTED
    er = bad.copy()
    oracle.relax(er)
    engine.solve_multi(bad, devices=[0, 0])                        # rates only: any values
    assert_bits_equal(bad, er, "rates-only, negative entry")
    with pytest.raises(engine.FwxError):
        engine.solve_multi(rate.copy(), devices=[0, 7])            # no such device here


def test_n4096_partitioned_whole_solve():
    n = 4096
    rate, nxt, _ = synth.make("d1", n, np.float32, seed=synth.BASE_SEED + 7)
    er, en = rate.copy(), nxt.copy()
    eu = oracle.relax_mt(er, en)
    for parts in (2, 8):
        gr, gn = rate.copy(), nxt.copy()
        u = engine.solve_multi(gr, gn, devices=[0] * parts, count_updates=True)
        assert_bits_equal(gr, er, "rate P=%d" % parts)
        assert_bits_equal(gn, en, "next P=%d" % parts)
        assert u == eu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_per_k_engine_on_partitions(parts, dtype):
    """FWX_ENGINE_PERK on a partitioned handle -- BASELINE config 4's shape: one launch per pivot and
    partition, pivot rows read from the exchanged 64-row snapshot panel, the look-ahead rows by one
    fused launch.  rate / next / hops and U against the oracle, ragged sizes included."""
    for kind, n in (("d1", 512), ("t1", 300), ("t2", 257), ("d2", 1000)):
        rate, nxt, hops = synth.make(kind, n, dtype, seed=parts * 100 + n)
        er, en, eh = rate.copy(), nxt.copy(), hops.copy()
        eu = oracle.relax(er, en, eh)
        gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
        u = engine.solve_multi(gr, gn, gh, devices=[0] * parts, engine=engine.FWX_ENGINE_PERK,
                               count_updates=True)
        assert_bits_equal(gr, er, "rate %s n=%d P=%d" % (kind, n, parts))
        assert_bits_equal(gn, en, "next %s n=%d P=%d" % (kind, n, parts))
        assert_bits_equal(gh, eh, "hops %s n=%d P=%d" % (kind, n, parts))
        assert u == eu
        gr = rate.copy()
        engine.solve_multi(gr, devices=[0] * parts, engine=engine.FWX_ENGINE_PERK)
        assert_bits_equal(gr, er, "rates-only %s n=%d P=%d" % (kind, n, parts))
    with engine.DeviceMatrix(128, dtype, with_next=True, devices=[0, 0]) as dm:
        dm.enable_path_log()                       # the per-k kernel keeps no trace on slabs
        rate, nxt, _ = synth.make("d1", 128, dtype, seed=1)
        dm.upload(rate, nxt)
        with pytest.raises(engine.FwxError) as e:
            dm.solve(engine=engine.FWX_ENGINE_PERK)
        assert e.value.status == FWX_ERR_UNSUPPORTED


@pytest.mark.parametrize("eng", [engine.FWX_ENGINE_FUSED, engine.FWX_ENGINE_PERK])
def test_pivot_ranges_on_a_partitioned_handle(eng):
    """A solve over [k_begin, k_end) is resumable on partitions as on one device (no trace):
    three ragged ranges in a row equal the oracle's whole solve, and each stage its prefix."""
    n = 700
    rate, nxt, hops = synth.make("t1", n, np.float32, seed=12)
    with engine.DeviceMatrix(n, np.float32, with_next=True, with_hops=True, devices=[0, 0, 0]) as dm:
        dm.upload(rate, nxt, hops)
        er, en, eh = rate.copy(), nxt.copy(), hops.copy()
        for k0, k1 in ((0, 37), (37, 411), (411, n)):
            oracle.relax(er, en, eh, k0, k1)
            dm.solve(engine=eng, k_begin=k0, k_end=k1)
            gr, gn, gh = dm.download()
            assert_bits_equal(gr, er, "rate after [%d,%d)" % (k0, k1))
            assert_bits_equal(gn, en, "next after [%d,%d)" % (k0, k1))
            assert_bits_equal(gh, eh, "hops after [%d,%d)" % (k0, k1))


def test_exception_barrier_and_no_leak_on_the_multi_path():
    """fwx.h: no C++ exception crosses the ABI.  fwx_test_fail_after arms a bad_alloc at the next
    internal allocation point: the partitioned solve must come back as FWX_ERR_OOM -- through
    fwx_matrix_solve on a handle and through the one-shot entry point, whose handle must be destroyed
    on the way out (free HBM returns to where it was) -- and the next call must work."""
    from floydwarshall_amd import _lib, hip
    n = 1024
    rate, nxt, _ = synth.make("d1", n, np.float32, seed=5)
    er, en, _ = _expect(rate, nxt)
    L = _lib.lib()
    with engine.DeviceMatrix(n, np.float32, with_next=True, devices=[0, 0]) as dm:
        dm.upload(rate, nxt)
        L.fwx_test_fail_after(1)
        with pytest.raises(engine.FwxError) as e:
            dm.solve()
        assert e.value.status == _lib.FWX_ERR_OOM
        dm.upload(rate, nxt)
        dm.solve()
        assert_bits_equal(dm.download()[0], er, "handle solve after the injected failure")
    engine.solve_multi(rate.copy(), nxt.copy(), devices=[0, 0, 0])       # warm: pool, RCCL-free path
    hip.synchronize()
    free0, _ = hip.mem_get_info()
    for countdown in (1, 2):               # 1: inside the solve; 2: after the download, before parking
        L.fwx_test_fail_after(countdown)
        with pytest.raises(engine.FwxError) as e:
            engine.solve_multi(rate.copy(), nxt.copy(), devices=[0, 0, 0])
        assert e.value.status == _lib.FWX_ERR_OOM
    L.fwx_test_fail_after(0)
    hip.synchronize()
    free1, _ = hip.mem_get_info()
    assert free1 >= free0, "the failed one-shot calls leaked %d bytes of HBM" % (free0 - free1)
    gr, gn = rate.copy(), nxt.copy()
    engine.solve_multi(gr, gn, devices=[0, 0, 0])
    assert_bits_equal(gr, er, "one-shot after the injected failures")
    assert_bits_equal(gn, en, "next")


def test_one_shot_calls_reuse_their_handle():
    """fwx_solve_multi_* parks its handle (slabs, streams, events, communicator) keyed by shape and
    device list; the second call with the same key must not pay create + destroy again, and calls
    with other keys in between must not confuse the pool."""
    import time
    n = 2048
    rate, nxt, _ = synth.make("d2", n, np.float64, seed=8)
    er, en, _ = _expect(rate, nxt)
    times = []
    for i in range(4):
        gr, gn = rate.copy(), nxt.copy()
        t0 = time.perf_counter()
        engine.solve_multi(gr, gn, devices=[0, 0])
        times.append(time.perf_counter() - t0)
        assert_bits_equal(gr, er, "rate, call %d" % i)
        assert_bits_equal(gn, en, "next, call %d" % i)
        if i == 1:                         # another key in between: different shape and field set
            small, _, _ = synth.make("d1", 256, np.float32, seed=3)
            es = small.copy()
            oracle.relax(es)
            engine.solve_multi(small, devices=[0, 0, 0])
            assert_bits_equal(small, es, "the other key")
    # a handle solve of the same matrix, device resident upload/download included, for scale
    with engine.DeviceMatrix(n, np.float64, with_next=True, devices=[0, 0]) as dm:
        t0 = time.perf_counter()
        dm.upload(rate, nxt)
        dm.solve()
        dm.download()
        t_handle = time.perf_counter() - t0
    assert min(times[1:]) < 1.5 * t_handle, (times, t_handle)


def test_config4_n16384_p8_partitioned_equals_the_whole_oracle_solve():
    """BASELINE config 4 at its own size: N = 16384 f32, P = 8 row partitions (logical: one GPU here),
    rates + next-hops through fwx_solve_multi_f32, on both engines -- against the committed digests
    of the WHOLE CPU-oracle solve of this matrix (tests/golden/config4_n16384_digests.json, made by
    tests/golden/make_config4_digests.py: 270 s + 205 s of oracle on the GPU box's host cores)."""
    from helpers import digest, load_golden
    gold = load_golden("config4_n16384_digests.json")
    n = gold["n"]
    rate, nxt = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
    for eng in (engine.FWX_ENGINE_FUSED, engine.FWX_ENGINE_PERK):
        gr, gn = rate.copy(), nxt.copy()
        u = engine.solve_multi(gr, gn, devices=[0] * 8, engine=eng, count_updates=True)
        assert digest(gr) == gold["rate_digest"], "rates, engine %d" % eng
        assert digest(gn) == gold["next_digest"], "next-hops, engine %d" % eng
        assert u == gold["U"]
    gr = rate.copy()
    engine.solve_multi(gr, devices=[0] * 8)                     # rates only: max-form kernels on slabs
    assert digest(gr) == gold["rate_digest"]


def test_config5_n32768_p8_partitioned_with_next_hops_and_path_lengths():
    """BASELINE config 5 as SURVEY.md 8d states it: N = 32768 f32, rates + next + hops, P = 8 row
    partitions (logical).  Three 256-pivot stretches pinned to the oracle (first pivots, across the middle
    partition boundary, last pivots: helpers.config5_solve_with_oracle_slices -- each two 128-pivot
    launches of the pair schedule); monotonicity; 10^6 sampled best-rate walks; then the same matrix
    through a traced partitioned handle: same bits, and the reference's exact `_path` list for EVERY
    sampled pair whose walk length differs from `hops` (helpers.check_walks_and_exact_lists)."""
    from helpers import check_walks_and_exact_lists, config5_solve_with_oracle_slices, dev, host
    n, P = 32768, 8
    rate0, next0 = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 4)
    hops0 = (next0 >= 0).astype(np.int32)
    with engine.DeviceMatrix(n, np.float32, with_next=True, with_hops=True, devices=[0] * P) as dm:
        dm.upload(rate0, next0, hops0)
        rate, nxt, hops = config5_solve_with_oracle_slices(dm, n)
    del hops0
    assert bool((rate >= rate0).all())
    rnd = np.random.default_rng(7)
    src = rnd.integers(0, n, 1000000).astype(np.int32)
    dst = rnd.integers(0, n, 1000000).astype(np.int32)
    d_next, d_rate0 = dev(nxt), dev(rate0)
    ln, prod, _ = engine.dev_follow_paths(d_next, dev(src), dev(dst), edge_rate_t=d_rate0)
    ln, prod = host(ln), host(prod)
    del d_next, d_rate0
    # the traced, partitioned solve of the same input (one call, whole range): same bits; exact lists
    # through the slab-local trace
    with engine.DeviceMatrix(n, np.float32, with_next=True, devices=[0] * P) as dm:
        dm.enable_path_log()
        dm.upload(rate0, next0)
        dm.solve()
        tr, tn, _ = dm.download()
        assert_bits_equal(tr, rate, "traced partitioned rates vs the ranged solve")
        assert_bits_equal(tn, nxt, "traced partitioned next-hops")
        del tr, tn
        differ = check_walks_and_exact_lists(rate0, rate, nxt, hops, src, dst, ln, prod,
                                             lambda a, b: dm.query_exact_batch(a, b, cap=256))
    assert differ < 0.02 * len(src)          # (walk and list almost always agree; not a property, a sanity check)


def _distinct_devices():
    return list(range(min(engine.device_count(), 8)))


@pytest.mark.skipif("engine.device_count() < 2", reason="needs two or more MI355X in one process")
@pytest.mark.parametrize("exchange", ["rccl", "peer"])
def test_distinct_devices_equal_the_oracle(exchange):
    """What the one-GPU boxes cannot run: one partition per REAL device, the snapshot panels
    travelling over xGMI -- by grouped ncclBroadcast on an ncclCommInitAll communicator of all the
    devices (what AUTO picks), and by peer copies.  Both engines, rate + next + hops + U against the
    oracle, the path trace against the list-faithful restatement, a ragged order, and the one-shot
    entry point twice (the second call from the handle pool)."""
    devs = _distinct_devices()
    xchg = engine.FWX_XCHG_RCCL if exchange == "rccl" else engine.FWX_XCHG_PEER
    for kind, n, dtype in (("d1", 1024, np.float32), ("t1", 517, np.float64)):
        rate, nxt, hops = synth.make(kind, n, dtype, seed=n)
        er, en, eh = rate.copy(), nxt.copy(), hops.copy()
        eu = oracle.relax(er, en, eh)
        for eng in (engine.FWX_ENGINE_FUSED, engine.FWX_ENGINE_PERK):
            with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True, devices=devs, exchange=xchg) as dm:
                assert dm.parts() == (len(devs), xchg)
                if exchange == "rccl":
                    assert dm.comm_ranks() == len(devs)
                dm.upload(rate, nxt, hops)
                u = dm.solve(engine=eng, count_updates=True)
                gr, gn, gh = dm.download()
            assert_bits_equal(gr, er, "rate %s n=%d engine=%d" % (kind, n, eng))
            assert_bits_equal(gn, en, "next")
            assert_bits_equal(gh, eh, "hops")
            assert u == eu
        for _ in range(2):
            gr, gn = rate.copy(), nxt.copy()
            engine.solve_multi(gr, gn, devices=devs, exchange=xchg)
            assert_bits_equal(gr, er, "one-shot rate")
            assert_bits_equal(gn, en, "one-shot next")
    n = 200
    rate, nxt, _ = synth.make("t1", n, np.float64, seed=3)
    m = lf.run_algo(lf.from_dense([("X", "C%03d" % i) for i in range(n)], rate, nxt), np.float64)
    paths = lf.path_indices(m)
    with engine.DeviceMatrix(n, np.float64, with_next=True, devices=devs, exchange=xchg) as dm:
        dm.enable_path_log()
        dm.upload(rate, nxt)
        dm.solve()
        rnd = np.random.default_rng(9)
        src = rnd.integers(0, n, 200).astype(np.int32)
        dst = rnd.integers(0, n, 200).astype(np.int32)
        got = dm.query_exact_batch(src, dst)
        for q in range(len(src)):
            assert tuple(got[q]) == paths[src[q]][dst[q]]


@pytest.mark.skipif("engine.device_count() < 2", reason="needs two or more MI355X in one process")
def test_distinct_devices_config4_digests():
    """BASELINE config 4 as it is meant (N = 16384 f32 over all the devices of the node, RCCL) against
    the committed whole-oracle digests, per-k engine (the bench's) and fused."""
    from helpers import digest, load_golden
    gold = load_golden("config4_n16384_digests.json")
    n = gold["n"]
    rate, nxt = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
    devs = _distinct_devices()
    for eng in (engine.FWX_ENGINE_PERK, engine.FWX_ENGINE_FUSED):
        gr, gn = rate.copy(), nxt.copy()
        engine.solve_multi(gr, gn, devices=devs, engine=eng)
        assert digest(gr) == gold["rate_digest"]
        assert digest(gn) == gold["next_digest"]
