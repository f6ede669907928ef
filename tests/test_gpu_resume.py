"""f3, exactly: resume instead of re-solve (fwx_matrix_enable_resume / fwx_matrix_resolve).

The reference throws the solved matrix away on every accepted price change
(/root/reference/src/lib/ProcessRequests.hs:99-102) and runs runAlgo from pivot 0
(:82-84 -> Algorithms.hs:19-20).  A changed input entry (i,j) is an operand only of steps i and j
(Algorithms.hs:58-60), so the engine restarts at the last stored state before min(i,j), with just the
changed entries replayed up to there.  The bar is the reference's: every result -- rates, next-hops,
path lengths, exact `_path` lists, through any sequence of changes -- must equal a from-scratch solve
of the changed input, bit for bit."""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, host, synth
from floydwarshall_amd._lib import FWX_ERR_INVALID, FWX_ERR_UNSUPPORTED

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def _fresh_lists(n, dtype, rate, nxt, src, dst):
    """Exact `_path` lists of a from-scratch traced solve (per-k engine: a different kernel family)."""
    with engine.DeviceMatrix(n, dtype, with_next=True) as dm:
        dm.enable_path_log()
        dm.upload(rate, nxt)
        dm.solve(engine=engine.FWX_ENGINE_PERK)
        return dm.query_exact_batch(src, dst, cap=16 * n)


@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0, 0]], ids=["one-device", "P2", "P4"])
@pytest.mark.parametrize("dtype,with_hops,traced", [(np.float64, True, True), (np.float32, False, True),
                                                    (np.float32, True, False), (np.float64, False, False)])
def test_resolve_equals_a_from_scratch_solve(dtype, with_hops, traced, devices):
    """devices: the same on a ROW-PARTITIONED handle (round 4): checkpoints and all-pivot panels per slab,
    every partition keeps the exchanged pivot rows, the changed entries are replayed on the partition
    that owns their row."""
    n, cps = 512, 3
    rnd = np.random.default_rng(77)
    rate, nxt, hops = synth.make("d2", n, dtype, seed=5)
    with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=with_hops, devices=devices) as dm:
        if traced:
            dm.enable_path_log()
        with pytest.raises(engine.FwxError) as e:
            dm.enable_resume(cps)                              # replays start from the KEPT input
        assert e.value.status == FWX_ERR_INVALID
        dm.keep_input()
        assert dm.enable_resume(cps) == cps                    # checkpoints at 128, 256, 384
        dm.upload(rate, nxt, hops if with_hops else None)
        dm.solve()
        cur_r, cur_n, cur_h = rate.copy(), nxt.copy(), hops.copy()
        plan = [(400, 500), (130, 300), (300, 131), (40, 90), (255, 256), (256, 257), (510, 511),
                (384, 385), (127, 128), (128, 129)]
        for step, (u, v) in enumerate(plan):
            # one accepted price change: entries (u,v) and (v,u) (Algorithms.hs:36-37)
            idx = np.array([u * n + v, v * n + u], dtype=np.int64)
            if step == 4:
                vals = np.array([0.0, 0.0], dtype=dtype)       # the pair stops trading: "no route" entries
                nv, hv = np.array([-1, -1], dtype=np.int32), np.array([0, 0], dtype=np.int32)
            else:
                # a quote moves a little, downwards: D2's potential keeps every cycle product below 1
                # (prices with arbitrage make the reference's lists revisit vertices and explode)
                vals = (rate.reshape(-1)[idx] * (0.9 + 0.1 * rnd.random(2))).astype(dtype)
                nv, hv = np.array([v, u], dtype=np.int32), np.array([1, 1], dtype=np.int32)
            if step == 7:                                      # several entries at once
                idx = np.concatenate([idx, [450 * n + 451, 390 * n + 500]])
                vals = np.concatenate([vals, (rate.reshape(-1)[idx[2:]] * 0.95).astype(dtype)])
                nv = np.concatenate([nv, np.array([451, 500], dtype=np.int32)])
                hv = np.concatenate([hv, np.array([1, 1], dtype=np.int32)])
            cur_r.reshape(-1)[idx] = vals
            cur_n.reshape(-1)[idx] = nv
            cur_h.reshape(-1)[idx] = hv
            started = dm.resolve(idx, vals, nv, hv if with_hops else None)
            lowest = int(min(min(i // n, i % n) for i in idx))
            assert started == max([p for p in (0, 128, 256, 384) if p <= lowest]), (step, started, lowest)
            er, en, eh = cur_r.copy(), cur_n.copy(), cur_h.copy()
            oracle.relax(er, en, eh)                           # runAlgo 0 on the changed input
            gr, gn, gh = dm.download()
            assert_bits_equal(gr, er, "rate after change %d (resumed at %d)" % (step, started))
            assert_bits_equal(gn, en, "next after change %d" % step)
            if with_hops:
                assert_bits_equal(gh, eh, "hops after change %d" % step)
            if traced and step % 3 == 0:
                src = rnd.integers(0, n, 200).astype(np.int32)
                dst = rnd.integers(0, n, 200).astype(np.int32)
                src[:4], dst[:4] = (u, v, u, 5), (v, u, 7, v)
                assert dm.query_exact_batch(src, dst, cap=16 * n) == _fresh_lists(n, dtype, cur_r, cur_n, src, dst)
        if devices is not None:
            return          # (a partitioned handle refuses matrices outside the domain with next-hops)
        # outside the reference's domain nothing is resumed (the per-k engine runs the full solve) ...
        idx = np.array([300 * n + 301], dtype=np.int64)
        bad = np.array([-0.5], dtype=dtype)
        cur_r.reshape(-1)[idx] = bad
        assert dm.resolve(idx, bad, np.array([301], dtype=np.int32), np.array([1], dtype=np.int32)
                          if with_hops else None) == 0
        er, en, eh = cur_r.copy(), cur_n.copy(), cur_h.copy()
        oracle.relax(er, en, eh)
        gr, gn, _ = dm.download()
        assert_bits_equal(gr, er, "rate, patch outside the domain")
        assert_bits_equal(gn, en, "next, patch outside the domain")
        # ... and once the input is back inside, the next full solve records again and resuming returns
        good = rate.reshape(-1)[idx].astype(dtype)
        cur_r.reshape(-1)[idx] = good
        assert dm.resolve(idx, good, np.array([301], dtype=np.int32), np.array([1], dtype=np.int32)
                          if with_hops else None) == 0
        idx2 = np.array([500 * n + 420], dtype=np.int64)
        v2 = (rate.reshape(-1)[idx2] * 0.97).astype(dtype)
        cur_r.reshape(-1)[idx2] = v2
        cur_n.reshape(-1)[idx2] = 420
        cur_h.reshape(-1)[idx2] = 1
        assert dm.resolve(idx2, v2, np.array([420], dtype=np.int32), np.array([1], dtype=np.int32)
                          if with_hops else None) == 384
        er, en, eh = cur_r.copy(), cur_n.copy(), cur_h.copy()
        oracle.relax(er, en, eh)
        gr, gn, gh = dm.download()
        assert_bits_equal(gr, er, "rate, after resuming again")
        assert_bits_equal(gn, en, "next, after resuming again")
        if with_hops:
            assert_bits_equal(gh, eh, "hops, after resuming again")


def test_what_invalidates_a_recording():
    """Only the solve OF THE KEPT INPUT can be resumed: a second solve on top of the first, a plain
    patch_input, a new upload, counting U or another engine all lead to a full solve (resumed at 0),
    which records afresh."""
    n = 256
    rate, nxt, _ = synth.make("d1", n, np.float32, seed=9)
    idx = np.array([200 * n + 201], dtype=np.int64)
    nv = np.array([201], dtype=np.int32)

    def change(dm, val, **kw):
        v = np.array([val], dtype=np.float32)
        rate.reshape(-1)[idx] = v
        started = dm.resolve(idx, v, nv, **kw)
        er, en = rate.copy(), nxt.copy()
        oracle.relax(er, en)
        gr, gn, _ = dm.download()
        assert_bits_equal(gr, er, "rate")
        assert_bits_equal(gn, en, "next")
        return started

    with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
        dm.keep_input()
        assert dm.enable_resume(1) == 1                        # one checkpoint, at pivot 128
        dm.upload(rate, nxt)
        dm.solve()
        assert change(dm, 0.41) == 128
        dm.solve()                                             # solved twice over: not the input's solve
        assert change(dm, 0.42) == 0
        assert change(dm, 0.43) == 128                         # ... which recorded afresh
        dm.patch_input(idx, np.array([0.44], dtype=np.float32), nv)
        rate.reshape(-1)[idx] = np.float32(0.44)
        dm.solve()
        assert change(dm, 0.45) == 128                         # patch_input + full solve record too
        assert change(dm, 0.46, count_updates=True) == 0       # U of a resumed solve would be partial
        assert change(dm, 0.47, engine=engine.FWX_ENGINE_PERK) == 0
        assert change(dm, 0.48) == 0                           # the per-k solve recorded nothing
        assert change(dm, 0.49) == 128
        dm.upload(rate, nxt)
        assert change(dm, 0.50) == 0                           # a new upload: solved from scratch
    with engine.DeviceMatrix(256, np.float32, with_next=True) as dm:    # the trace must be enabled FIRST
        dm.keep_input()
        dm.enable_resume(2)
        with pytest.raises(engine.FwxError) as e:
            dm.enable_path_log()
        assert e.value.status == FWX_ERR_INVALID
    with engine.DeviceMatrix(64, np.float32, with_next=True) as dm:     # AUTO solves this in one launch: no passes
        dm.keep_input()
        with pytest.raises(engine.FwxError) as e:
            dm.enable_resume(2)
        assert e.value.status == FWX_ERR_UNSUPPORTED
    for odd_n, dtype in ((130, np.float32), (255, np.float64)):         # odd orders: the handle pads its rows
        with engine.DeviceMatrix(odd_n, dtype, with_next=True) as dm:
            dm.keep_input()
            assert dm.enable_resume(2) >= 1
    # partitions are cut on multiples of 64 once n >= 128 P (512 rows over 3: 0, 128, 320), so every checkpoint
    # is a block start ...
    rate, nxt, _ = synth.make("d1", 512, np.float32, seed=10)
    with engine.DeviceMatrix(512, np.float32, with_next=True, devices=[0, 0, 0]) as dm:
        assert [dm.part_rows(p) for p in range(3)] == [(0, 128), (128, 192), (320, 192)]
        dm.keep_input()
        assert dm.enable_resume(3) == 3
        dm.upload(rate, nxt)
        dm.solve()
        idx = np.array([400 * 512 + 300], dtype=np.int64)
        v = (rate.reshape(-1)[idx] * np.float32(0.9)).astype(np.float32)
        rate.reshape(-1)[idx] = v
        assert dm.resolve(idx, v, np.array([300], dtype=np.int32)) == 256
        er, en = rate.copy(), nxt.copy()
        oracle.relax(er, en)
        gr, gn, _ = dm.download()
        assert_bits_equal(gr, er, "rate")
        assert_bits_equal(gn, en, "next")
    # ... below that the partitions are balanced and unaligned (300 rows over 3: 0, 100, 200) and a checkpoint
    # must still be a block start: only pivot 64 (inside partition 0) qualifies -- and it works
    rate, nxt, _ = synth.make("d1", 300, np.float32, seed=11)
    with engine.DeviceMatrix(300, np.float32, with_next=True, devices=[0, 0, 0]) as dm:
        assert [dm.part_rows(p)[0] for p in range(3)] == [0, 100, 200]
        dm.keep_input()
        assert dm.enable_resume(3) == 1
        dm.upload(rate, nxt)
        dm.solve()
        idx = np.array([250 * 300 + 150], dtype=np.int64)
        v = (rate.reshape(-1)[idx] * np.float32(0.9)).astype(np.float32)
        rate.reshape(-1)[idx] = v
        assert dm.resolve(idx, v, np.array([150], dtype=np.int32)) == 64
        er, en = rate.copy(), nxt.copy()
        oracle.relax(er, en)
        gr, gn, _ = dm.download()
        assert_bits_equal(gr, er, "rate")
        assert_bits_equal(gn, en, "next")


@pytest.mark.parametrize("devices", [None, [0, 0, 0, 0]], ids=["one-device", "P4"])
def test_session_resumes_after_price_changes_and_answers_like_a_fresh_session(devices):
    """(devices: the session behind `fwx_cli --devices`, the resident matrix row-partitioned -- it resumes
    like the single-device one.)  The AppState trigger on top (Types.hs:35-37, ProcessRequests.hs:82-85): 32 exchanges x 8
    currencies = 256 vertices; a feed of price changes between known vertices, a best-rate request
    after each.  The session's re-solves resume at a checkpoint whenever the changed vertices allow;
    every answer (rate and the reference's exact `_path`) equals that of a session that never resumes
    and of a fresh session fed the same rates."""
    rnd = np.random.default_rng(4)
    ccys = ["C%d" % i for i in range(8)]
    price = dict(zip(ccys, 0.5 + 1.5 * rnd.random(len(ccys))))
    log = []

    def quote(exch, a, b, t):
        return (t, exch, a, b, price[b] / price[a] * (0.97 + 0.03 * rnd.random()),
                price[a] / price[b] * (0.97 + 0.03 * rnd.random()))

    t = 1000
    for e in range(32):
        for i in range(len(ccys)):
            log.append(quote("E%02d" % e, ccys[i], ccys[(i + 1) % len(ccys)], t))   # a ring: every ccy appears
            if rnd.random() < 0.5:
                log.append(quote("E%02d" % e, ccys[i], ccys[(i + 3) % len(ccys)], t))
    s, plain = host.Session(device=0), host.Session(device=0)
    plain.set_checkpoints(0)
    if devices is not None:
        s.set_devices(devices, min_vertices=0)
    for r in log:
        assert s.update_rates(*r) and plain.update_rates(*r)
    vs = sorted({(r[1], c) for r in log for c in (r[2], r[3])})
    assert len(vs) == 256
    a, b = vs[3], vs[200]
    assert s.find_best_rate(a, b) == plain.find_best_rate(a, b)
    for step in range(24):
        t += 1
        old = log[int(rnd.integers(0, len(log)))]
        log.append(quote(old[1], old[2], old[3], t))
        assert s.update_rates(*log[-1]) and plain.update_rates(*log[-1])
        for _ in range(6):
            a, b = (vs[int(x)] for x in rnd.integers(0, len(vs), 2))
            try:
                want = plain.find_best_rate(a, b)
            except host.AlgoError as err:
                with pytest.raises(host.AlgoError) as e2:
                    s.find_best_rate(a, b)
                assert str(e2.value) == str(err)
                continue
            assert s.find_best_rate(a, b) == want
    assert s.solves == plain.solves == 25 and s.patched_solves == 24
    assert plain.resumed_solves == 0 and s.parts == (1 if devices is None else len(devices))
    assert s.resumed_solves >= 12, s.resumed_solves            # changes below the first checkpoint cannot
    assert s.resumed_pivots >= 32 * s.resumed_solves
    fresh = host.Session(device=0)
    for r in log:
        fresh.update_rates(*r)
    r1, n1, h1 = s.solved_matrix()
    r2, n2, h2 = fresh.solved_matrix()
    assert_bits_equal(r1, r2, "rate")
    assert np.array_equal(n1, n2) and np.array_equal(h1, h2)
