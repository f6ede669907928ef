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
