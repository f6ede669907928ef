"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle, bit for bit.

Bar (task section 3): bit-exact for next-hop / hops indices; rates are compared bit-exact too (the
north star allows 1e-12 rel, the per-k kernel has no reason to differ by a single ulp).
"""
import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth
from oracle import list_faithful as lf

from helpers import (assert_bits_equal, dev, dev_empty, dev_sync, dev_zeros, golden_dense,
                     golden_rates_dict, host, host_cat, load_golden, path_from_trace)

pytestmark = pytest.mark.gpu


def _solve_and_compare(rate, nxt, hops, **kw):
    er = rate.copy()
    en = None if nxt is None else nxt.copy()
    eh = None if hops is None else hops.copy()
    eu = oracle.relax(er, en, eh, kw.get("k_begin", 0), kw.get("k_end") or None)
    gr = rate.copy()
    gn = None if nxt is None else nxt.copy()
    gh = None if hops is None else hops.copy()
    u = engine.solve(gr, gn, gh, count_updates=True, **kw)
    assert_bits_equal(gr, er, "rate")
    if nxt is not None:
        assert_bits_equal(gn, en, "next")
    if hops is not None:
        assert_bits_equal(gh, eh, "hops")
    assert u == eu
    return gr, gn, gh


def test_device_present_and_library_loaded():
    assert engine.device_count() >= 1


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_golden_4x4_on_gpu(dtype):
    # /root/reference/src/test/AlgorithmsTest.hs:66-77 through the HIP path
    g = load_golden("algorithms_4x4.json")
    rate, nxt, hops, _ = golden_dense(g["initial"], dtype)
    orate, onext, ohops = rate.copy(), nxt.copy(), hops.copy()
    engine.solve(rate, nxt, hops)
    erate, enext, ehops, epaths = golden_dense(g["solved"], dtype)
    if dtype == np.float64:
        assert_bits_equal(rate, erate, "solved rate")          # the reference's own numbers
    else:
        # the reference has no f32 mode: its f64 golden rounds to within 1 ulp of the f32 loop,
        # and the f32 loop itself is pinned bit for bit by the oracle run at f32
        assert np.allclose(rate, erate, rtol=1e-6)
        oracle.relax(orate, onext, ohops)
        assert_bits_equal(rate, orate, "f32 solved rate vs the f32 oracle")
        assert np.array_equal(nxt, onext) and np.array_equal(hops, ohops)
    assert np.array_equal(nxt, enext) and np.array_equal(hops, ehops)
    for i in range(4):
        for j in range(4):
            assert tuple(engine.follow_path(nxt, i, j)) == epaths[i][j]


def test_empty_and_tiny():
    engine.solve(np.zeros((0, 0)))
    for n in (1, 2, 3):
        rate, nxt, hops = synth.make("d1", n, np.float64, seed=n)
        _solve_and_compare(rate, nxt, hops)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [5, 63, 64, 65, 127, 200, 256, 257, 511, 1000])
def test_ragged_sizes_all_fields(n, dtype):
    rate, nxt, hops = synth.make("d1", n, dtype, seed=1000 + n)
    _solve_and_compare(rate, nxt, hops)


@pytest.mark.parametrize("kind", ["d1", "d2", "t1", "t2", "t3"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_distributions(kind, dtype):
    rate, nxt, hops = synth.make(kind, 384, dtype, seed=77)
    _solve_and_compare(rate, nxt, hops)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rates_only_and_next_only(dtype):
    rate, nxt, _ = synth.make("d2", 320, dtype, seed=3)
    _solve_and_compare(rate, None, None)
    _solve_and_compare(rate, nxt, None)


def test_config2_n1024_fp64_dense_random():
    """BASELINE.json configs[1]: N=1024 dense random fp64, per-k kernel, bit-exact."""
    for kind in ("d1", "d2"):
        rate, nxt, hops = synth.make(kind, 1024, np.float64, seed=synth.BASE_SEED + 1)
        _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_PERK)


def test_wide_strip_config_n2048_and_8192_prefix():
    # n >= 2048 and n >= 8192 select other launch configurations; check both against the oracle
    rate, nxt, _ = synth.make("d1", 2048, np.float32, seed=11)
    _solve_and_compare(rate, nxt, None, k_begin=0, k_end=96)
    rate, _, _ = synth.make("d1", 8192, np.float32, seed=12)
    _solve_and_compare(rate, None, None, k_begin=4000, k_end=4012)


def test_serpentine_off_is_identical():
    rate, nxt, hops = synth.make("d1", 300, np.float32, seed=5)
    a = _solve_and_compare(rate, nxt, hops, serpentine=True)
    b = _solve_and_compare(rate, nxt, hops, serpentine=False)
    assert_bits_equal(a[0], b[0])


def test_k_range_resume():
    rate, nxt, hops = synth.make("d2", 200, np.float64, seed=21)
    full = _solve_and_compare(rate, nxt, hops)
    r, n_, h = rate.copy(), nxt.copy(), hops.copy()
    engine.solve(r, n_, h, k_begin=0, k_end=77)
    engine.solve(r, n_, h, k_begin=77, k_end=200)
    assert_bits_equal(r, full[0])
    assert np.array_equal(n_, full[1]) and np.array_equal(h, full[2])


def test_gpu_matches_list_faithful_paths():
    """Whole `_path` lists (Algorithms.hs:55) from the GPU's next-hops, no-arbitrage input."""
    n = 40
    rate, nxt, hops = synth.make("d2", n, np.float64, seed=8)
    vertices = [("X", "C%03d" % i) for i in range(n)]
    m = lf.run_algo(lf.from_dense(vertices, rate, nxt))
    paths = lf.path_indices(m)
    engine.solve(rate, nxt, hops)
    for i in range(n):
        for j in range(n):
            assert tuple(engine.follow_path(nxt, i, j)) == paths[i][j]
            assert hops[i, j] == len(paths[i][j])
            assert rate[i, j] == m[i][j][0]


def test_device_matrix_handle_and_query():
    g = load_golden("algorithms_4x4.json")
    rate, nxt, hops, _ = golden_dense(g["initial"])
    m = engine.DeviceMatrix(4, np.float64, with_next=True, with_hops=True)
    m.upload(rate, nxt, hops)
    assert m.solve(count_updates=True) == 8
    erate, enext, ehops, epaths = golden_dense(g["solved"])
    r, n_, h = m.download()
    assert_bits_equal(r, erate)
    assert np.array_equal(n_, enext) and np.array_equal(h, ehops)
    for i in range(4):
        for j in range(4):
            q_rate, q_path = m.query(i, j)
            assert q_rate == erate[i, j] and tuple(q_path) == epaths[i][j]
    m.close()


def test_device_step_api_and_partition_emulation():
    """fwx_dev_relax / fwx_dev_panel on caller-owned device memory: P logical row partitions on ONE
    GPU (owner panel -> D2D copy standing in for the RCCL broadcast -> everyone relaxes its slab with
    the snapshot panel) must equal the single-slab solve and the oracle bit for bit."""
    n, P, B = 512, 4, 32
    rate, nxt, _ = synth.make("d1", n, np.float32, seed=31)
    er, en = rate.copy(), nxt.copy()
    eu = oracle.relax(er, en)

    # (a) single slab, pivots read in place
    r1, n1 = dev(rate), dev(nxt)
    upd = dev_zeros((engine.FWX_UPDATE_SHARDS,), np.int64)
    engine.dev_relax(r1, n, 0, 0, n, next_t=n1, updates_t=upd)
    assert_bits_equal(host(r1), er, "single slab rate")
    assert_bits_equal(host(n1), en, "single slab next")
    assert int(host(upd).sum()) == eu

    # (b) P partitions, snapshot panels
    rows = n // P
    slabs = [dev(rate[p * rows:(p + 1) * rows]) for p in range(P)]
    nslabs = [dev(nxt[p * rows:(p + 1) * rows]) for p in range(P)]
    for k0 in range(0, n, B):
        owner, off = k0 // rows, k0 % rows
        w = dev_empty((B, n), np.float32)
        engine.dev_panel(slabs[owner][off:off + B], n, k0, w, next_t=nslabs[owner][off:off + B])
        for p in range(P):
            wp = w.clone()                      # stands in for the broadcast
            if p != owner:
                engine.dev_relax(slabs[p], n, p * rows, k0, k0 + B, pivots_t=wp, next_t=nslabs[p])
            else:
                if off > 0:
                    engine.dev_relax(slabs[p][:off], n, p * rows, k0, k0 + B, pivots_t=wp,
                                     next_t=nslabs[p][:off])
                if off + B < rows:
                    engine.dev_relax(slabs[p][off + B:], n, p * rows + off + B, k0, k0 + B,
                                     pivots_t=wp, next_t=nslabs[p][off + B:])
    assert_bits_equal(host_cat(slabs), er, "partitioned rate")
    assert_bits_equal(host_cat(nslabs), en, "partitioned next")


def test_full_size_properties_n4096_fp32():
    """Properties at a size the oracle cannot finish quickly:
      * monotonicity: no rate decreases (Algorithms.hs:55 only ever replaces by a larger value);
      * rate == product of the input edge rates along the next-hop path (to fp32 rounding);
      * a second solve moves no rate by more than rounding (exact idempotence does NOT hold in
        floating point: re-associated products can win by an ulp);
      * oracle parity on k-slices taken from the middle of the GPU solve, bit for bit."""
    n = 4096
    rate0, nxt0, _ = synth.make("d2", n, np.float32, seed=synth.BASE_SEED + 3)
    rate, nxt = rate0.copy(), nxt0.copy()
    u1 = engine.solve(rate, nxt, count_updates=True)
    assert u1 > 0
    assert np.all(rate >= rate0)
    r2 = rate.copy()
    engine.solve(r2)
    assert np.all(r2 >= rate)
    assert float(np.max((r2 - rate) / rate.clip(min=1e-30))) < 1e-5
    rnd = np.random.default_rng(1)
    for _ in range(300):
        s, d = (int(x) for x in rnd.integers(0, n, 2))
        if s == d:
            continue
        path = engine.follow_path(nxt, s, d)
        assert path and path[-1] == d
        prod, cur = 1.0, s
        for v in path:
            prod *= float(rate0[cur, v])
            cur = v
        assert abs(prod - float(rate[s, d])) <= 1e-5 * float(rate[s, d])
    # mid-solve slices: GPU state at k=k0, then pivots [k0,k0+6) on both sides
    for k0 in (0, 1531, n - 6):
        r, nx = rate0.copy(), nxt0.copy()
        if k0:
            engine.solve(r, nx, k_begin=0, k_end=k0)
        _solve_and_compare(r, nx, None, k_begin=k0, k_end=k0 + 6)


def test_config4_n16384_fp32_k_slices_vs_oracle():
    """BASELINE.json configs[3] size (1 GiB matrix, wide-strip launch configuration): oracle parity
    on pivot slices at the start and in the middle of the solve; serpentine on == off."""
    n = 16384
    rate0, _ = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
    _solve_and_compare(rate0, None, None, k_begin=0, k_end=3)
    r = rate0.copy()
    engine.solve(r, k_begin=0, k_end=301)
    a = _solve_and_compare(r, None, None, k_begin=301, k_end=304)
    b = r.copy()
    engine.solve(b, k_begin=301, k_end=304, serpentine=False)
    assert_bits_equal(a[0], b, "serpentine off")


# ---------------------------------------------------------------------------------------------
# Fused engine (64 pivots per pass): must be bit-identical to the per-k engine and the oracle
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [4, 8, 60, 64, 68, 128, 132, 200, 256, 388, 512, 1000])
def test_fused_engine_ragged_sizes(n, dtype):
    rate, nxt, _ = synth.make("d1", n, dtype, seed=2000 + n)
    _solve_and_compare(rate, nxt, None, engine=engine.FWX_ENGINE_FUSED)
    _solve_and_compare(rate, None, None, engine=engine.FWX_ENGINE_FUSED)


@pytest.mark.parametrize("kind", ["d1", "d2", "t1", "t2", "t3"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fused_engine_distributions(kind, dtype):
    """Ties (earliest pivot must win), sparse/unreachable, inf/NaN/negative inputs."""
    rate, nxt, _ = synth.make(kind, 320, dtype, seed=78)
    _solve_and_compare(rate, nxt, None, engine=engine.FWX_ENGINE_FUSED)


def test_fused_engine_golden_4x4():
    g = load_golden("algorithms_4x4.json")
    rate, nxt, _, _ = golden_dense(g["initial"])
    engine.solve(rate, nxt, engine=engine.FWX_ENGINE_FUSED)
    erate, enext, _, epaths = golden_dense(g["solved"])
    assert_bits_equal(rate, erate, "solved rate")
    assert np.array_equal(nxt, enext)


def test_fused_engine_k_range_and_unsupported():
    rate, nxt, hops = synth.make("d2", 300, np.float64, seed=22)
    _solve_and_compare(rate, nxt, None, engine=engine.FWX_ENGINE_FUSED, k_begin=37, k_end=211)
    # hops ride through the fused engine's panels: whole range and pivot ranges alike
    _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_FUSED)
    _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_FUSED, k_begin=37, k_end=211)
    # n not a multiple of the 16-byte vector width: the device-pointer API refuses the fused engine
    # (it cannot pad memory it does not own) ...
    odd, _, _ = synth.make("d1", 63, np.float32, seed=1)
    with pytest.raises(engine.FwxError):
        engine.dev_solve(dev(odd), engine=engine.FWX_ENGINE_FUSED)
    # ... a handle owns its arrays and pads them (next test but one)
    want = odd.copy()
    oracle.relax(want)
    with engine.DeviceMatrix(63, np.float32, with_next=False) as dm:
        dm.upload(odd)
        dm.solve(engine=engine.FWX_ENGINE_FUSED)
        got, _, _ = dm.download()
    assert_bits_equal(got, want, "rate")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [3, 63, 257, 301, 511, 1001])
def test_host_api_pads_odd_sizes_for_the_fused_engine(n, dtype):
    """... while fwx_solve_* pads an odd-sized matrix on the device with +0.0 rows / columns
    (inert: never a pivot, never improved), so AUTO keeps the fused engine for any n >= 256 and an
    explicit FUSED request works for every n.  Same bits, same U, all input kinds."""
    for kind in ("d2", "t1", "t3"):
        rate, nxt, _ = synth.make(kind, n, dtype, seed=3000 + n)
        _solve_and_compare(rate, nxt, None, engine=engine.FWX_ENGINE_FUSED)
        _solve_and_compare(rate, None, None, engine=engine.FWX_ENGINE_FUSED)
        _solve_and_compare(rate, nxt, None)                                     # AUTO
        _solve_and_compare(rate, None, None, k_begin=n // 3, k_end=n - 1)       # AUTO, pivot range


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [243, 1017, 3101])
def test_single_device_handles_pad_their_rows_for_any_order(n, dtype):
    """buildMatrix (Algorithms.hs:29) produces any n; `fwx_matrix_create` keeps the arrays at a device
    order rounded up to 16 bytes of rate elements (inert padding: never a pivot, a +0.0 target never
    improves), so a plain handle runs the FUSED engine, keeps hops and the path trace, answers queries
    and RESUMES at every order -- nothing falls back to one launch per pivot.  Everything against the
    oracle (rates, next, hops, U) and, for the `_path` lists, against their defining properties."""
    kind = "d2" if n != 1017 else "t1"                      # 1017: tie-heavy, earliest pivot must win
    rate, nxt, hops = synth.make(kind, n, dtype, seed=4300 + n)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    eu = oracle.relax_mt(er, en, hops=eh, fast=True)
    it = np.uint64 if dtype == np.float64 else np.uint32
    # rates only / + next / + next + hops, FUSED and AUTO and PERK, with U
    for fields in (0, 1, 2):
        for eng in (engine.FWX_ENGINE_FUSED, engine.FWX_ENGINE_AUTO, engine.FWX_ENGINE_PERK):
            if eng == engine.FWX_ENGINE_PERK and (fields != 2 or n > 1100):
                continue
            with engine.DeviceMatrix(n, dtype, with_next=fields >= 1, with_hops=fields == 2) as dm:
                dm.upload(rate, nxt if fields >= 1 else None, hops if fields == 2 else None)
                u = dm.solve(engine=eng, count_updates=True)
                gr, gn, gh = dm.download()
                assert u == eu, (fields, eng)
                assert np.array_equal(gr.view(it), er.view(it)), (fields, eng)
                if fields >= 1:
                    assert np.array_equal(gn, en), (fields, eng)
                    s_, d_ = n - 1, n // 2                   # a query in the last (odd) row
                    r_, path = dm.query(s_, d_)
                    assert path == [int(x) for x in engine.follow_path(en, s_, d_)]
                    assert np.asarray(r_, dtype=dtype).view(it) == er[s_, d_].view(it)
                if fields == 2:
                    assert np.array_equal(gh, eh), (fields, eng)
                # uncounted: the max-form kernels (and the double pass where the order reaches it)
                dm.upload(rate, nxt if fields >= 1 else None, hops if fields == 2 else None)
                dm.solve(engine=eng)
                gr2, gn2, gh2 = dm.download()
                assert np.array_equal(gr2.view(it), er.view(it)), (fields, eng, "max form")
                if fields >= 1:
                    assert np.array_equal(gn2, en)
                if fields == 2:
                    assert np.array_equal(gh2, eh)
    # path trace + kept input + resume: a change of two entries late in the matrix resumes at a checkpoint
    with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True) as dm:
        dm.enable_path_log()
        dm.keep_input()
        placed = dm.enable_resume(3)
        assert placed == 3
        dm.upload(rate, nxt, hops)
        dm.solve()
        gr, gn, gh = dm.download()
        assert np.array_equal(gr.view(it), er.view(it)) and np.array_equal(gn, en) and np.array_equal(gh, eh)
        rnd = np.random.default_rng(n)
        pairs = rnd.integers(0, n, size=(64, 2)).astype(np.int32)
        pairs[0] = (n - 1, 0)
        pairs[1] = (0, n - 1)
        lists = dm.query_exact_batch(pairs[:, 0], pairs[:, 1], cap=4 * n)
        for q, (a, b) in enumerate(pairs):
            assert len(lists[q]) == eh[a, b], (a, b)         # length of the reference's list = hops
            if lists[q]:
                assert lists[q][-1] == b and lists[q][0] == en[a, b]
        i, j = n - 2, n - 1                                  # both indices beyond the last checkpoint
        r2 = rate.copy()
        idx = np.array([i * n + j, j * n + i], dtype=np.int64)
        vals = (r2.reshape(-1)[idx] * dtype(0.9)).astype(dtype)
        r2.reshape(-1)[idx] = vals
        started = dm.resolve(idx, vals, np.array([j, i], dtype=np.int32), np.array([1, 1], dtype=np.int32))
        assert started > 0 and started % 64 == 0
        er2, en2, eh2 = r2.copy(), nxt.copy(), hops.copy()
        oracle.relax_mt(er2, en2, hops=eh2, fast=True)
        gr, gn, gh = dm.download()
        assert np.array_equal(gr.view(it), er2.view(it)) and np.array_equal(gn, en2) and np.array_equal(gh, eh2)
        lists = dm.query_exact_batch(pairs[:, 0], pairs[:, 1], cap=4 * n)
        for q, (a, b) in enumerate(pairs):
            assert len(lists[q]) == eh2[a, b], (a, b)
        # patch_input (full solve of the patched kept input) takes the caller's n x n indices too
        r3 = r2.copy()
        idx = np.array([(n - 1) * n + 1], dtype=np.int64)
        vals = (r3.reshape(-1)[idx] * dtype(0.5)).astype(dtype)
        r3.reshape(-1)[idx] = vals
        dm.patch_input(idx, vals, np.array([1], dtype=np.int32), np.array([1], dtype=np.int32))
        dm.solve()
        er3, en3, eh3 = r3.copy(), nxt.copy(), hops.copy()
        oracle.relax_mt(er3, en3, hops=eh3, fast=True)
        gr, gn, gh = dm.download()
        assert np.array_equal(gr.view(it), er3.view(it)) and np.array_equal(gn, en3) and np.array_equal(gh, eh3)


def test_config3_n8192_fp32_fused_vs_perk_and_oracle_slices():
    """BASELINE.json configs[2]: N=8192 fp32 blocked (LDS-tiled).  Full solve with the fused
    engine == full solve with the per-k engine, bit for bit (rates and next-hops); oracle parity on
    pivot slices taken mid-solve."""
    n = 8192
    rate0, nxt0 = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 2)
    a_r, a_n = rate0.copy(), nxt0.copy()
    ua = engine.solve(a_r, a_n, engine=engine.FWX_ENGINE_FUSED, count_updates=True)
    b_r, b_n = rate0.copy(), nxt0.copy()
    ub = engine.solve(b_r, b_n, engine=engine.FWX_ENGINE_PERK, count_updates=True)
    assert ua == ub
    assert_bits_equal(a_r, b_r, "fused vs per-k rate")
    assert_bits_equal(a_n, b_n, "fused vs per-k next")
    r, nx = rate0.copy(), nxt0.copy()
    engine.solve(r, nx, engine=engine.FWX_ENGINE_FUSED, k_begin=0, k_end=1000)
    _solve_and_compare(r, nx, None, engine=engine.FWX_ENGINE_FUSED, k_begin=1000, k_end=1003)
    _solve_and_compare(rate0, nxt0, None, engine=engine.FWX_ENGINE_FUSED, k_begin=0, k_end=2)


@pytest.mark.parametrize("with_extras", [False, True])
def test_fused_device_api_partition_emulation(with_extras):
    """fwx_dev_panel_snap + fwx_dev_relax_fused on P logical partitions of one GPU: rate + next, and
    (with_extras) hops -- which travel with the panel -- and the path trace, kept slab-local."""
    n, P = 640, 3
    rate, nxt, hops = synth.make("t1", n, np.float32, seed=33)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    eu = oracle.relax(er, en, eh)
    bounds = [n * p // P for p in range(P + 1)]
    cut = lambda a, p: dev(a[bounds[p]:bounds[p + 1]])  # noqa: E731
    slabs = [cut(rate, p) for p in range(P)]
    nslabs = [cut(nxt, p) for p in range(P)]
    hslabs = [cut(hops, p) if with_extras else None for p in range(P)]
    traces = [engine.Trace(bounds[p + 1] - bounds[p], n) if with_extras else None for p in range(P)]
    wss = [engine.FusedWorkspace(n, bounds[p + 1] - bounds[p], np.float32, with_next=True,
                                 with_hops=with_extras) for p in range(P)]
    upd = dev_zeros((engine.FWX_UPDATE_SHARDS,), np.int64)
    B = engine.FWX_FUSED_BLOCK
    for owner in range(P):
        k0 = bounds[owner]
        while k0 < bounds[owner + 1]:
            k1 = min(k0 + B, bounds[owner + 1])
            lo, hi = k0 - bounds[owner], k1 - bounds[owner]
            w = wss[owner].w[0][:k1 - k0]
            wh = wss[owner].wh[0][:k1 - k0] if with_extras else None
            engine.dev_panel_snap(slabs[owner][lo:hi], n, k0, w, block_next_t=nslabs[owner][lo:hi],
                                  block_hops_t=hslabs[owner][lo:hi] if with_extras else None, w_hops_t=wh,
                                  trace=traces[owner].rows(lo, hi) if with_extras else None)
            for p in range(P):
                wp = w.clone()                           # stands in for the broadcast
                whp = wh.clone() if with_extras else None
                # counting keeps the compare-form kernel; the uncounted variant below takes the
                # max-form + arg re-scan kernel
                engine.dev_relax_fused(slabs[p], n, bounds[p], k0, k1, wp, wss[p], next_t=nslabs[p],
                                       hops_t=hslabs[p], wh_t=whp, trace=traces[p], updates_t=upd)
            k0 = k1
    assert_bits_equal(host_cat(slabs), er, "partitioned fused rate")
    assert_bits_equal(host_cat(nslabs), en, "partitioned fused next")
    assert int(host(upd).sum()) == eu
    if with_extras:
        assert_bits_equal(host_cat(hslabs), eh, "partitioned fused hops")
        # the slab-local traces, stacked, are the single-device trace: check `last` against a
        # single-device traced solve of the same input
        with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt)
            dm.solve(engine=engine.FWX_ENGINE_FUSED)
            src = np.arange(0, n, 7, dtype=np.int32)
            dst = ((src * 13 + 5) % n).astype(np.int32)
            want = dm.query_exact_batch(src, dst)
        last, at_col, at_row = (host_cat([getattr(t, f) for t in traces])
                                for f in ("last", "at_col", "at_row"))
        for q in range(len(src)):
            assert path_from_trace(last, at_col, at_row, nxt, int(src[q]), int(dst[q])) == want[q]


def test_batch_path_follow_matches_host_walk():
    n = 300
    rate0, nxt0, hops0 = synth.make("d2", n, np.float64, seed=61)
    rate, nxt, hops = rate0.copy(), nxt0.copy(), hops0.copy()
    engine.solve(rate, nxt, hops)
    rnd = np.random.default_rng(3)
    src = rnd.integers(0, n, 5000).astype(np.int32)
    dst = rnd.integers(0, n, 5000).astype(np.int32)
    ln, prod, paths = engine.dev_follow_paths(dev(nxt), dev(src), dev(dst), edge_rate_t=dev(rate0),
                                              path_cap=n)
    ln, prod, paths = host(ln), host(prod), host(paths)
    for q in range(len(src)):
        exp = oracle.follow_path(nxt, int(src[q]), int(dst[q]))
        assert ln[q] == len(exp) == hops[src[q], dst[q]]
        assert list(paths[q, :ln[q]]) == exp
        if ln[q]:
            assert abs(prod[q] - rate[src[q], dst[q]]) <= 1e-12 * rate[src[q], dst[q]]
    # unreachable pairs and cycles
    sparse_r, sparse_n, _ = synth.make("t2", 64, np.float64, seed=2)
    engine.solve(sparse_r, sparse_n)
    s2 = np.arange(64, dtype=np.int32)
    ln2, _, _ = engine.dev_follow_paths(dev(sparse_n), dev(s2), dev(s2))
    assert (host(ln2) == 0).all()                             # src == dst: empty path
    loop = np.array([[-1, 1, 1], [0, -1, 0], [0, 0, -1]], dtype=np.int32)
    ln3, _, _ = engine.dev_follow_paths(dev(loop), dev(np.array([0], dtype=np.int32)),
                                        dev(np.array([2], dtype=np.int32)))
    assert int(host(ln3)[0]) == -5


def test_config5_n32768_fp32_with_next_hop_matrix():
    """BASELINE.json configs[4] size on ONE GPU: N = 32768 fp32 with the predecessor (next-hop) matrix and
    path lengths, through a plain handle (fused engine; two passes per main launch).  Three 256-pivot
    stretches pinned to the oracle (helpers.config5_solve_with_oracle_slices: first pivots, middle, last
    pivots -- rate, next AND hops), then full best-rate path reconstruction for 10^6 sampled (src, dst)
    pairs on the device and the exact `_path` lists of a traced solve of the same input
    (helpers.check_walks_and_exact_lists)."""
    from helpers import check_walks_and_exact_lists, config5_solve_with_oracle_slices
    n = 32768
    rate_h, next_h = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 4)
    hops_h = (next_h >= 0).astype(np.int32)
    with engine.DeviceMatrix(n, np.float32, with_next=True, with_hops=True) as dm:
        dm.upload(rate_h, next_h, hops_h)
        solved_h, nxt_h, hops = config5_solve_with_oracle_slices(dm, n)
    del hops_h
    assert bool((solved_h >= rate_h).all())
    rnd = np.random.default_rng(7)
    src = rnd.integers(0, n, 1000000).astype(np.int32)
    dst = rnd.integers(0, n, 1000000).astype(np.int32)
    rate0, nxt = dev(rate_h), dev(nxt_h)
    ln, prod, paths = engine.dev_follow_paths(nxt, dev(src), dev(dst), edge_rate_t=rate0, path_cap=4)
    ln, prod, paths = host(ln), host(prod), host(paths)
    del rate0, nxt
    assert int(ln.max()) < 64
    short = (ln >= 1) & (ln <= 4)
    last = paths[short, ln[short] - 1]
    assert bool((last == dst[short]).all())
    with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
        dm.enable_path_log()
        dm.upload(rate_h, next_h)
        dm.solve()
        tr, tn, _ = dm.download()
        assert_bits_equal(tr, solved_h, "traced rates vs the ranged solve")
        assert_bits_equal(tn, nxt_h, "traced next-hops vs the ranged solve")
        del tr, tn
        check_walks_and_exact_lists(rate_h, solved_h, nxt_h, hops, src, dst, ln, prod,
                                    lambda a, b: dm.query_exact_batch(a, b, cap=256))


# ---------------------------------------------------------------------------------------------
# max-form kernel (v_pk_mul_f32 + v_max3_f32): only taken for f32, rates only, no update counting,
# after the domain check (all entries >= +0, no NaN).  Must still be bit-identical.
# ---------------------------------------------------------------------------------------------

def _rates_only_fused(rate):
    exp = rate.copy()
    oracle.relax(exp)
    got = rate.copy()
    engine.solve(got, engine=engine.FWX_ENGINE_FUSED)          # no next, no counting
    assert_bits_equal(got, exp, "max-form rate")
    return got


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["d1", "d2", "t1", "t2", "t4"])
@pytest.mark.parametrize("n", [64, 132, 516])
def test_max_form_kernel_inside_its_domain(kind, n, dtype):
    rate, _, _ = synth.make(kind, n, dtype, seed=300 + n)
    assert (rate >= 0).all() and not np.isnan(rate).any() and not np.signbit(rate).any()
    got = _rates_only_fused(rate)
    if kind == "t4":
        assert np.isinf(got).any()                              # the overflow edge was exercised


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("poison", ["nan", "negative", "negzero"])
def test_max_form_kernel_is_not_taken_outside_its_domain(poison, dtype):
    rate, _, _ = synth.make("d1", 260, dtype, seed=9)
    rate[17, 201] = {"nan": np.nan, "negative": -0.75, "negzero": -0.0}[poison]
    _rates_only_fused(rate)
    rate, _, _ = synth.make("t3", 260, dtype, seed=10)
    _rates_only_fused(rate)


def test_domain_check_and_flagged_device_api():
    n = 384
    rate, _, _ = synth.make("t4", n, np.float32, seed=12)
    exp = rate.copy()
    oracle.relax(exp)
    r = dev(rate)
    assert engine.dev_check_nonneg(r, n)
    engine.dev_solve_fused(r, n)                               # takes the max-form kernel
    assert_bits_equal(host(r), exp, "max-form via device API")
    bad = rate.copy()
    bad[5, 7] = np.nan
    assert not engine.dev_check_nonneg(dev(bad), n)
    bad[5, 7] = -0.0
    assert not engine.dev_check_nonneg(dev(bad), n)
    r64 = rate.astype(np.float64)
    assert engine.dev_check_nonneg(dev(r64), n)                 # f64 has a max form too
    r64[9, 1] = -1.0
    assert not engine.dev_check_nonneg(dev(r64), n)


def test_max_form_full_size_n8192_vs_perk():
    n = 8192
    rate0, _ = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 2)
    a = rate0.copy()
    engine.solve(a, engine=engine.FWX_ENGINE_FUSED)
    b = rate0.copy()
    engine.solve(b, engine=engine.FWX_ENGINE_PERK)
    assert_bits_equal(a, b, "max-form fused vs per-k, N=8192")


@pytest.mark.parametrize("with_next", [False, True])
def test_dev_solve_blocking_api_with_lookahead(with_next):
    """fwx_dev_solve on caller-owned device memory: the fused engine's look-ahead schedule (side
    stream) against the oracle, odd pivot ranges included."""
    n = 708
    rate, nxt, _ = synth.make("t1", n, np.float32, seed=91)
    er, en = rate.copy(), nxt.copy()
    eu = oracle.relax(er, en if with_next else None, None, 33, 650)
    r = dev(rate)
    nx = dev(nxt) if with_next else None
    u = engine.dev_solve(r, next_t=nx, engine=engine.FWX_ENGINE_FUSED, k_begin=33, k_end=650,
                         count_updates=True)
    assert u == eu
    assert_bits_equal(host(r), er, "rate")
    if with_next:
        assert_bits_equal(host(nx), en, "next")
    # the max-form path (no counting) through the same schedule
    r2 = dev(rate)
    engine.dev_solve(r2, engine=engine.FWX_ENGINE_FUSED, k_begin=33, k_end=650)
    assert_bits_equal(host(r2), er, "rate (max form)")


def test_config4_n16384_fp32_full_solve_fused_equals_perk():
    """BASELINE.json's headline size, whole solve: the per-k engine (16384 launches), the fused
    engine in max form (rates only) and the fused engine with the next-hop matrix (arg re-scan) must
    agree bit for bit on all 2^28 rates -- and with the WHOLE ORACLE SOLVE of this matrix, whose
    digests are committed under tests/golden/ (270 s of CPU, tools/full_parity_n16384.py);
    next-hops agree with a per-k + next run on a k-prefix.  (~10 s of GPU time.)"""
    from helpers import digest
    n = 16384
    gold = load_golden("config4_n16384_digests.json")
    rate_h, next_h = synth.d1_uniform(n, np.float32, synth.BASE_SEED + 3)
    r0 = dev(rate_h)
    a = r0.clone()
    engine.dev_solve(a, engine=engine.FWX_ENGINE_PERK)
    a_h = host(a)
    assert digest(a_h) == gold["rate_digest"], "per-k engine vs the whole oracle solve"
    b = r0.clone()
    engine.dev_solve(b, engine=engine.FWX_ENGINE_FUSED)                 # max form
    assert_bits_equal(host(b), a_h, "fused max form vs per-k")
    del b
    c = r0.clone()
    nc = dev(next_h)
    engine.dev_solve(c, next_t=nc, engine=engine.FWX_ENGINE_FUSED)      # arg re-scan + next
    assert_bits_equal(host(c), a_h, "fused with next-hops vs per-k")
    assert digest(host(nc)) == gold["next_digest"], "next-hops vs the whole oracle solve"
    del c
    # next-hops: per-k vs fused on the first 1024 pivots
    d = r0.clone()
    nd = dev(next_h)
    engine.dev_solve(d, next_t=nd, engine=engine.FWX_ENGINE_PERK, k_end=1024)
    e = r0.clone()
    ne = dev(next_h)
    engine.dev_solve(e, next_t=ne, engine=engine.FWX_ENGINE_FUSED, k_end=1024)
    assert_bits_equal(host(d), host(e), "rates after 1024 pivots")
    assert_bits_equal(host(nd), host(ne), "next-hops after 1024 pivots")
    del d, e, nd, ne
    # every path of the full solve ends at its destination, product of input edges == rate
    rnd = np.random.default_rng(11)
    src = rnd.integers(0, n, 200000).astype(np.int32)
    dst = rnd.integers(0, n, 200000).astype(np.int32)
    ln, prod, _ = engine.dev_follow_paths(nc, dev(src), dev(dst), edge_rate_t=r0)
    ln, prod = host(ln), host(prod)
    ok = src != dst
    assert bool((ln[ok] >= 1).all())
    solved = a_h[src, dst].astype(np.float64)
    assert float(((np.abs(prod - solved) / np.maximum(solved, 1e-30))[ok]).max()) < 2e-5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [1, 2, 4, 7, 31, 63, 64, 65, 72, 100, 120, 127, 128])
def test_small_solve_single_launch(n, dtype):
    """n <= 128 (the reference's own regime): AUTO solves in one single-workgroup launch (a 64- or
    a 128-wide register tile); must equal the oracle and the per-k engine bit for bit, all fields,
    all distributions, k-ranges."""
    for kind in ("d1", "t1", "t2", "t3"):
        rate, nxt, hops = synth.make(kind, n, dtype, seed=500 + n)
        a = _solve_and_compare(rate, nxt, hops)                                  # AUTO -> small
        b = _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_PERK)
        assert_bits_equal(a[0], b[0])
        _solve_and_compare(rate, None, None)
        _solve_and_compare(rate, nxt, None)
    if n >= 4:
        rate, nxt, hops = synth.make("d2", n, dtype, seed=9)
        _solve_and_compare(rate, nxt, hops, k_begin=1, k_end=n - 1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_subnormal_products_are_not_flushed(dtype):
    """The reference multiplies IEEE doubles with gradual underflow.  Rates small enough that every
    multi-hop product is SUBNORMAL (or underflows to +0): the device must neither flush inputs nor
    outputs to zero, in any engine (per-k, fused compare form, fused max form)."""
    n = 192
    rnd = np.random.default_rng(17)
    tiny = np.finfo(dtype).tiny                      # smallest normal
    scale = np.sqrt(float(tiny)) * 0.5               # every two-hop product is < tiny/4: subnormal
    rate = (scale * (0.05 + rnd.random((n, n)))).astype(dtype)
    rate[rnd.random((n, n)) < 0.3] *= dtype(1e-2)    # some deeper inside the subnormal range
    np.fill_diagonal(rate, 0)
    direct = rnd.random((n, n)) < 0.6                # knock out direct edges so that
    rate[~direct] = 0                                # subnormal two-hop products win
    np.fill_diagonal(rate, 0)
    nxt = np.where(rate > 0, np.arange(n, dtype=np.int32)[None, :], -1).astype(np.int32)
    exp_r, exp_n = rate.copy(), nxt.copy()
    oracle.relax(exp_r, exp_n)
    sub = (exp_r > 0) & (exp_r < tiny)
    assert sub.sum() > n                             # the case really exercises subnormals
    for eng in (engine.FWX_ENGINE_PERK, engine.FWX_ENGINE_FUSED):
        r, nx = rate.copy(), nxt.copy()
        engine.solve(r, nx, engine=eng)
        assert_bits_equal(r, exp_r, "rate engine %d" % eng)
        assert_bits_equal(nx, exp_n, "next engine %d" % eng)
    r = rate.copy()
    engine.solve(r, engine=engine.FWX_ENGINE_FUSED)  # f32: max form
    assert_bits_equal(r, exp_r, "rate, rates-only fused")


def test_concurrent_solves_from_several_host_threads():
    """The boundary must be callable from any OS thread (GHC `safe` FFI calls migrate between OS
    threads): three threads solve three different matrices at once, twice each."""
    import threading
    results, errors = {}, []

    def work(tid):
        try:
            for rep in range(2):
                kind, n, dt = [("d1", 260, np.float32), ("t1", 200, np.float64), ("d2", 520, np.float32)][tid]
                rate, nxt, _ = synth.make(kind, n, dt, seed=900 + tid + 10 * rep)
                er, en = rate.copy(), nxt.copy()
                oracle.relax(er, en)
                engine.solve(rate, nxt)
                results[(tid, rep)] = (np.array_equal(rate.view(np.uint8), er.view(np.uint8))
                                       and np.array_equal(nxt, en))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(results) == 6 and all(results.values())


def test_solves_on_different_host_threads_overlap_on_the_device():
    """Every handle runs on its own non-blocking stream (never the legacy null stream), so solves
    issued from different host threads overlap on the device.  n = 128 solves are ONE single-
    workgroup launch each (1 of 256 CUs busy): four threads with a handle each must get clearly more
    solves per second than one thread alone -- on a shared blocking stream the launches would
    queue behind one another and the aggregate rate could not rise."""
    import threading
    import time
    n, reps = 128, 300
    rate, nxt, hops = synth.make("d2", n, np.float64, seed=12)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    oracle.relax(er, en, eh)

    def run(handle, out, idx):
        ok = True
        for _ in range(reps):
            handle.upload(rate, nxt, hops)
            handle.solve()
        r, nx, hp = handle.download()
        ok = ok and np.array_equal(r.view(np.uint64), er.view(np.uint64)) and np.array_equal(nx, en) \
            and np.array_equal(hp, eh)
        out[idx] = ok

    handles = [engine.DeviceMatrix(n, np.float64, with_next=True, with_hops=True) for _ in range(4)]
    try:
        out = [None] * 4
        run(handles[0], out, 0)                       # warm-up
        t0 = time.perf_counter()
        run(handles[0], out, 0)
        t_one = time.perf_counter() - t0
        ts = [threading.Thread(target=run, args=(handles[i], out, i)) for i in range(4)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        t_four = time.perf_counter() - t0
    finally:
        for h in handles:
            h.close()
    assert all(out)
    rate_one, rate_four = reps / t_one, 4 * reps / t_four
    assert rate_four > 1.5 * rate_one, (rate_one, rate_four)


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_kept_input_and_patch(devices, dtype):
    """fwx_matrix_keep_input / fwx_matrix_patch_input: a few entries of the kept input are replaced
    on the device and the (full) solve of the patched input equals the solve of the same matrix
    uploaded whole -- single-device and partitioned handles, with next + hops and the path trace."""
    n = 300
    rate, nxt, hops = synth.make("d2", n, dtype, seed=8)
    kw = dict(devices=devices) if devices else dict(device=0)
    with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True, **kw) as dm:
        dm.enable_path_log()
        with pytest.raises(engine.FwxError):
            dm.patch_input([5], [1.0])                          # nothing kept yet
        dm.keep_input()
        with pytest.raises(engine.FwxError):
            dm.patch_input([5], [1.0])                          # kept, but no upload yet
        dm.upload(rate, nxt, hops)
        dm.solve()
        rnd = np.random.default_rng(4)
        for step in range(3):
            idx = rnd.integers(0, n * n, 6).astype(np.int64)
            idx = idx[(idx // n) != (idx % n)]
            # (only ever lower a quote: the market stays free of arbitrage and the lists stay short)
            vals = (rate.flat[idx] * (0.9 + 0.1 * rnd.random(len(idx)))).astype(dtype)
            if step == 1:
                vals[0] = 0                                     # an edge disappears: no route, no path
            nv = np.where(vals != 0, idx % n, -1).astype(np.int32)
            hv = (vals != 0).astype(np.int32)
            rate.flat[idx], nxt.flat[idx], hops.flat[idx] = vals, nv, hv
            dm.patch_input(idx, vals, nv, hv)
            u = dm.solve(count_updates=True)
            er, en, eh = rate.copy(), nxt.copy(), hops.copy()
            eu = oracle.relax(er, en, eh)
            gr, gn, gh = dm.download()
            assert_bits_equal(gr, er, "rate after patch %d" % step)
            assert_bits_equal(gn, en, "next after patch %d" % step)
            assert_bits_equal(gh, eh, "hops after patch %d" % step)
            assert u == eu
            # the trace (and its next0) follow the patched input
            with engine.DeviceMatrix(n, dtype, with_next=True) as ref:
                ref.enable_path_log()
                ref.upload(rate, nxt)
                ref.solve()
                src = rnd.integers(0, n, 40).astype(np.int32)
                dst = rnd.integers(0, n, 40).astype(np.int32)
                assert dm.query_exact_batch(src, dst) == ref.query_exact_batch(src, dst)
        with pytest.raises(engine.FwxError):
            dm.patch_input([n * n], [1.0])                      # index out of range


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_patches_that_leave_the_domain_are_rechecked(dtype):
    """A handle remembers the domain check's answer for its arrays; a patch with a value outside the
    domain (negative, NaN, -0, or a non-zero rate patched in without its next-hop) must send the
    next solve through the check again -- and to the engine that handles such inputs."""
    n = 260
    rate, nxt, hops = synth.make("d1", n, dtype, seed=31)
    with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True, device=0) as dm:
        dm.keep_input()
        dm.upload(rate, nxt, hops)
        dm.solve()                                              # inside the domain: remembered
        cases = [(np.array([7 * n + 9]), dtype(-0.75)), (np.array([3 * n + 200]), dtype(np.nan)),
                 (np.array([11 * n + 2]), dtype(-0.0)), (np.array([5 * n + 6]), dtype(0.5))]
        for step, (idx, val) in enumerate(cases):
            idx = idx.astype(np.int64)
            vals = np.array([val], dtype=dtype)
            nv = np.array([idx[0] % n], dtype=np.int32)
            hv = np.array([1], dtype=np.int32)
            rate.flat[idx], nxt.flat[idx], hops.flat[idx] = vals, nv, hv
            dm.patch_input(idx, vals, nv, hv)
            dm.solve()
            er, en, eh = rate.copy(), nxt.copy(), hops.copy()
            with np.errstate(all="ignore"):
                oracle.relax(er, en, eh)
            gr, gn, gh = dm.download()
            assert_bits_equal(gr, er, "rate, step %d" % step)
            assert_bits_equal(gn, en, "next, step %d" % step)
            assert_bits_equal(gh, eh, "hops, step %d" % step)
    # a non-zero rate patched over a missing edge WITHOUT its next-hop: a positive rate without a path
    rate, nxt, hops = synth.make("t2", n, dtype, seed=32)       # sparse: zeros with next = -1
    zi = int(np.flatnonzero((rate == 0) & (~np.eye(n, dtype=bool)))[0])
    with engine.DeviceMatrix(n, dtype, with_next=True, device=0) as dm:
        dm.keep_input()
        dm.upload(rate, nxt)
        dm.solve()
        rate.flat[zi] = dtype(0.5)
        dm.patch_input(np.array([zi], dtype=np.int64), np.array([0.5], dtype=dtype))
        dm.solve()
        er, en = rate.copy(), nxt.copy()
        oracle.relax(er, en)
        gr, gn, _ = dm.download()
        assert_bits_equal(gr, er, "rate, orphan patch")
        assert_bits_equal(gn, en, "next, orphan patch")


def test_caller_supplied_stream_and_v1_opts_struct():
    """fwx_opts.stream (ABI v2): the blocking call runs on the caller's stream, so work queued on it
    beforehand is ordered before the solve -- here the upload of the input itself, asynchronously on
    that stream -- and a caller built against the v1 struct (no stream fields, struct_size = 40)
    still works."""
    import ctypes
    from floydwarshall_amd import _lib, hip
    n = 512
    rate, nxt, _ = synth.make("d2", n, np.float32, seed=21)
    er, en = rate.copy(), nxt.copy()
    oracle.relax(er, en)
    st = hip.Stream()
    d_rate, d_next = hip.DeviceArray((n, n), np.float32), hip.DeviceArray((n, n), np.int32)
    d_rate.copy_from_host(rate, st)                 # async on st; the solve below must wait for it
    d_next.copy_from_host(nxt, st)
    engine.dev_solve(d_rate, next_t=d_next, stream=st)
    assert_bits_equal(d_rate.numpy(st), er, "rate on the caller's stream")
    assert_bits_equal(d_next.numpy(st), en, "next on the caller's stream")

    class OptsV1(ctypes.Structure):                 # fwx_opts as ABI version 1 declared it
        _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("engine", ctypes.c_int32),
                    ("k_begin", ctypes.c_int32), ("k_end", ctypes.c_int32), ("block", ctypes.c_int32),
                    ("serpentine", ctypes.c_int32), ("updates_out", ctypes.POINTER(ctypes.c_uint64))]
    o = OptsV1()
    o.struct_size, o.device = ctypes.sizeof(OptsV1), -1
    assert ctypes.sizeof(OptsV1) == 40
    r2, n2 = rate.copy(), nxt.copy()
    fn = _lib.lib().fwx_solve_f32
    rc = fn(n, r2.ctypes.data_as(ctypes.c_void_p), n2.ctypes.data_as(ctypes.c_void_p), None,
            ctypes.cast(ctypes.byref(o), ctypes.POINTER(_lib.FwxOpts)))
    assert rc == 0
    assert_bits_equal(r2, er, "rate through the v1 options struct")
    o.struct_size = 16                              # too small to be any version of the struct
    assert fn(n, r2.ctypes.data_as(ctypes.c_void_p), None, None,
              ctypes.cast(ctypes.byref(o), ctypes.POINTER(_lib.FwxOpts))) == _lib.FWX_ERR_INVALID


def test_index_math_beyond_2_to_31_elements():
    """N = 49152: 2.4e9 entries (9 GiB of f32) -- every offset must be computed in 64 bits.
    Per-k engine vs the oracle on two pivots; fused engine (both forms) vs per-k on 64 pivots."""
    n = 49152
    rnd = np.random.default_rng(4242)
    r0_h = np.empty((n, n), dtype=np.float32)
    for lo in range(0, n, 4096):                       # in strips: no 19 GiB float64 temporary
        r0_h[lo:lo + 4096] = rnd.random((4096, n), dtype=np.float32) * np.float32(0.95) + np.float32(0.05)
    np.fill_diagonal(r0_h, 0.0)
    r0 = dev(r0_h)
    k0 = n - 70                                        # pivots near the END: the largest offsets
    oracle.relax_mt(r0_h, None, k0, k0 + 2)            # r0_h becomes the expectation
    a = r0.clone()
    engine.dev_solve(a, engine=engine.FWX_ENGINE_PERK, k_begin=k0, k_end=k0 + 2)
    assert_bits_equal(host(a), r0_h, "per-k vs oracle, two pivots at the far end")
    del r0_h
    engine.dev_solve(a, engine=engine.FWX_ENGINE_PERK, k_begin=k0 + 2, k_end=k0 + 64)
    a_h = host(a)
    del a
    b = r0.clone()
    engine.dev_solve(b, engine=engine.FWX_ENGINE_FUSED, k_begin=k0, k_end=k0 + 64)      # max form
    assert_bits_equal(host(b), a_h, "fused max form vs per-k")
    b.copy_(r0)
    nb_h = np.empty((n, n), dtype=np.int32)
    nb_h[:] = np.arange(n, dtype=np.int32)[None, :]
    np.fill_diagonal(nb_h, -1)
    nb = dev(nb_h)
    del nb_h
    engine.dev_solve(b, next_t=nb, engine=engine.FWX_ENGINE_FUSED, k_begin=k0, k_end=k0 + 64)
    assert_bits_equal(host(b), a_h, "fused with next-hops vs per-k")
    # next-hops written at the far end of the matrix are pivots of the slice or the direct edge
    tail = host(nb[n - 8:n]).astype(np.int64)
    cols = np.arange(n)[None, :]
    ok = (tail == cols) | ((tail >= k0) & (tail < k0 + 64)) | (tail == -1)
    assert bool(ok.all())


# ---------------------------------------------------------------------------------------------
# Exact `_path` lists: path trace + fwx_matrix_query_exact
# ---------------------------------------------------------------------------------------------

def _market_rates(n_exch, n_ccy, seed, density=0.5):
    """The reference's own kind of graph: per-exchange quotes plus its built-in rate-1.0 edges
    between the same currency on two exchanges (Algorithms.hs:35) -- exact ties everywhere."""
    rnd = np.random.default_rng(seed)
    ccys = ["C%02d" % i for i in range(n_ccy)]
    price = dict(zip(ccys, 0.5 + 1.5 * rnd.random(n_ccy)))
    rates = {}
    for e in range(n_exch):
        exch = "X%02d" % e
        for i in range(n_ccy):
            for j in range(i + 1, n_ccy):
                if rnd.random() < density:
                    a, b = ccys[i], ccys[j]
                    rates[((exch, a), (exch, b))] = price[b] / price[a] * (0.97 + 0.03 * rnd.random())
                    rates[((exch, b), (exch, a))] = price[a] / price[b] * (0.97 + 0.03 * rnd.random())
    return rates


def _exact_paths_case(m0, dtype=np.float64, solve_engine=engine.FWX_ENGINE_AUTO):
    """m0: list-form initial matrix.  Solve with the list-faithful reference and with the logged
    GPU solve; every entry's `_path` must be identical."""
    ref = lf.run_algo(m0, dtype)
    ref_paths = lf.path_indices(ref)
    _, rate, nxt, hops = lf.to_dense(m0, dtype)
    n = rate.shape[0]
    _, erate, enext, ehops = lf.to_dense(ref, dtype)
    u = engine.solve(rate.copy(), nxt.copy(), hops.copy(), count_updates=True,
                     engine=engine.FWX_ENGINE_PERK)
    with_hops = solve_engine != engine.FWX_ENGINE_FUSED        # the fused engine carries no hops
    dm = engine.DeviceMatrix(n, dtype, with_next=True, with_hops=with_hops)
    dm.enable_path_log()
    dm.upload(rate, nxt, hops if with_hops else None)
    assert dm.solve(count_updates=True, engine=solve_engine) == u
    assert dm.path_log_count() == u
    r, nx, hp = dm.download()
    assert_bits_equal(r, erate, "rate")
    assert np.array_equal(nx, enext) and (hp is None or np.array_equal(hp, ehops))
    differs_from_walk = 0
    for i in range(n):
        for j in range(n):
            if ehops[i, j] > 4 * n:
                continue                                    # arbitrage blow-up: beyond the buffer
            q_rate, q_path = dm.query_exact(i, j)
            assert tuple(q_path) == ref_paths[i][j], (i, j)
            assert q_rate == erate[i, j] or (np.isnan(q_rate) and np.isnan(erate[i, j]))
            try:
                walk = engine.follow_path(nx, i, j)
            except engine.FwxError:
                walk = None
            differs_from_walk += walk != q_path
    dm.close()
    return differs_from_walk


def test_exact_path_lists_on_a_tie_heavy_market():
    """Market-like input: the reference's stored lists differ from the next-hop walk for some
    entries (equal-rate detours through its 1.0 edges); query_exact reproduces them all."""
    m0 = lf.build_matrix(_market_rates(8, 6, seed=23))
    assert 32 < len(m0) <= 64
    differs = _exact_paths_case(m0)
    assert differs > 0          # the case really exercises what the plain walk cannot give


@pytest.mark.parametrize("solve_engine", [engine.FWX_ENGINE_AUTO, engine.FWX_ENGINE_FUSED])
@pytest.mark.parametrize("kind", ["d1", "t1", "t2", "t3"])
def test_exact_path_lists_dense_kinds(kind, solve_engine):
    n = 20
    rate, nxt, _ = synth.make(kind, n, np.float64, seed=77)
    vertices = [("X", "C%03d" % i) for i in range(n)]
    _exact_paths_case(lf.from_dense(vertices, rate, nxt), solve_engine=solve_engine)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_exact_path_lists_fused_engine_three_passes(dtype):
    """n = 136 (three 64-pivot passes, ragged last pass and ragged tiles): the trace kept by
    fused_rowpanel / fused_colpanel / fused_main against the list-faithful restatement, ties and
    sparse inputs, every entry."""
    n = 136
    vertices = [("X", "C%03d" % i) for i in range(n)]
    for kind in ("t1", "t2"):
        rate, nxt, _ = synth.make(kind, n, dtype, seed=12)
        ref = lf.path_indices(lf.run_algo(lf.from_dense(vertices, rate, nxt), dtype))
        with engine.DeviceMatrix(n, dtype, with_next=True) as dm:
            dm.enable_path_log()
            dm.upload(rate, nxt)
            dm.solve(engine=engine.FWX_ENGINE_FUSED)
            for i in range(0, n, 3):
                for j in range(n):
                    assert tuple(dm.query_exact(i, j)[1]) == ref[i][j], (kind, i, j)


def test_path_trace_on_a_reused_handle():
    """One handle, several uploads of different matrices (few updates, many updates, few again):
    the trace of each solve stands alone; every list equals the reference's."""
    n = 136                                              # > 128: the per-k engine writes the log
    vertices = [("X", "C%03d" % i) for i in range(n)]
    refs = {}

    def check(dm, name, rate, nxt, hops):
        dm.upload(rate, nxt, hops)
        u = dm.solve(count_updates=True)
        er, en, eh = rate.copy(), nxt.copy(), hops.copy()
        assert u == oracle.relax(er, en, eh) == dm.path_log_count()
        r, nx, hp = dm.download()
        assert_bits_equal(r, er, "rate")
        assert np.array_equal(nx, en) and np.array_equal(hp, eh)
        if name not in refs:
            refs[name] = lf.path_indices(lf.run_algo(lf.from_dense(vertices, rate, nxt), np.float64))
        rnd = np.random.default_rng(u % 1000)
        for _ in range(300):
            i, j = (int(x) for x in rnd.integers(0, n, size=2))
            assert tuple(dm.query_exact(i, j)[1]) == refs[name][i][j]

    sparse = synth.make("t2", n, np.float64, seed=5)     # few updates
    dense = synth.make("t1", n, np.float64, seed=6)      # many more
    with engine.DeviceMatrix(n, np.float64, with_next=True, with_hops=True) as dm:
        dm.enable_path_log()
        check(dm, "sparse", *sparse)
        check(dm, "sparse", *sparse)
        check(dm, "dense", *dense)
        check(dm, "dense", *dense)
        check(dm, "sparse", *sparse)


def test_path_log_lifecycle():
    """query_exact needs a completed traced solve of the CURRENT upload; a traced solve starts from
    an uploaded input (solving the solved matrix again is refused); pivot ranges are refused for
    traced matrices."""
    n = 96
    rate, nxt, _ = synth.make("t1", n, np.float32, seed=5)
    want_r, want_n = rate.copy(), nxt.copy()
    oracle.relax(want_r, want_n)
    with engine.DeviceMatrix(n, np.float32, with_next=True) as dm:
        dm.enable_path_log()
        dm.upload(rate, nxt)
        with pytest.raises(engine.FwxError):
            dm.query_exact(0, 1)
        with pytest.raises(engine.FwxError):
            dm.solve(k_begin=0, k_end=n // 2)
        u1 = dm.solve(count_updates=True)
        assert u1 == dm.path_log_count()
        with pytest.raises(engine.FwxError):              # a traced solve needs a fresh upload
            dm.solve()
        r, nx = dm.download()[:2]
        assert_bits_equal(r, want_r, "rate")
        assert np.array_equal(nx, want_n)
        q_rate, q_path = dm.query_exact(3, 7)
        assert q_rate == want_r[3, 7] and q_path[-1] == 7
        dm.upload(rate, nxt)                              # new input: the old log is stale
        with pytest.raises(engine.FwxError):
            dm.query_exact(3, 7)


@pytest.mark.parametrize("solve_engine", [engine.FWX_ENGINE_AUTO, engine.FWX_ENGINE_PERK,
                                          engine.FWX_ENGINE_FUSED])
def test_exact_path_lists_80_vertices(solve_engine):
    """n = 80: AUTO keeps the path trace in the 128-wide single launch, PERK in relax_k, FUSED in
    the three kernels of a 64-pivot pass (two passes here); all must rebuild every list of the
    reference."""
    m0 = lf.build_matrix(_market_rates(10, 8, seed=23))
    assert 64 < len(m0) <= 80
    assert _exact_paths_case(m0, solve_engine=solve_engine) > 0


from hostile_inputs import hostile_matrix as _hostile_matrix  # noqa: E402


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fuzz_hostile_values_every_engine(dtype):
    """A few hundred small matrices of hostile values through every engine that accepts them
    (single launch 64- and 128-wide, per-k, fused compare form; rates only / + next / + hops)."""
    rnd = np.random.default_rng(4242 if dtype == np.float64 else 4343)
    sizes = [1, 2, 3, 4, 5, 8, 13, 16, 31, 32, 33, 48, 63, 64, 65, 66, 80, 100, 127, 128, 129, 132, 160]
    with np.errstate(all="ignore"):
        for rep in range(18):
            for n in sizes:
                rate, nxt, hops = _hostile_matrix(rnd, n, dtype)
                _solve_and_compare(rate, nxt, hops)                              # AUTO
                _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_PERK)
                variant = rep % 3
                if variant == 0:
                    _solve_and_compare(rate, None, None)
                elif variant == 1:
                    _solve_and_compare(rate, nxt, None)
                if n % (16 // np.dtype(dtype).itemsize) == 0:
                    _solve_and_compare(rate, nxt if variant else None, None,
                                       engine=engine.FWX_ENGINE_FUSED)
                if n >= 3:
                    k0 = int(rnd.integers(0, n - 1))
                    k1 = int(rnd.integers(k0 + 1, n + 1))
                    _solve_and_compare(rate, nxt, hops, k_begin=k0, k_end=k1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_per_k_relax_with_a_skipped_row_range(dtype):
    """fwx_dev_relax_skip: one launch per pivot over a slab, leaving a row range alone (the rows a
    look-ahead step has already relaxed).  Rows outside the range must equal the oracle's, rows
    inside must be untouched; a misaligned range is refused."""
    n, lo, hi, k0, k1 = 520, 128, 192, 200, 264
    rate, nxt, _ = synth.make("d2", n, dtype, seed=31)
    want_r, want_n = rate.copy(), nxt.copy()
    oracle.relax(want_r, want_n, None, k0, k1)
    r_t, n_t = dev(rate), dev(nxt)
    w = dev_empty((k1 - k0, n), dtype)                                # time-k snapshots of the pivots
    engine.dev_panel_snap(r_t[k0:k1], n, k0, w)
    engine.dev_relax(r_t, n, 0, k0, k1, pivots_t=w, next_t=n_t, skip=(lo, hi))
    got_r, got_n = host(r_t), host(n_t)
    keep = np.ones(n, dtype=bool)
    keep[lo:hi] = False
    assert_bits_equal(got_r[keep], want_r[keep], "rows outside the skipped range")
    assert np.array_equal(got_n[keep], want_n[keep])
    assert_bits_equal(got_r[lo:hi], rate[lo:hi], "skipped rows")
    assert np.array_equal(got_n[lo:hi], nxt[lo:hi])
    with pytest.raises(engine.FwxError):
        engine.dev_relax(r_t, n, 0, k0, k1, pivots_t=w, next_t=n_t, skip=(lo + 2, hi))


@pytest.mark.parametrize("dtype,n", [(np.float64, 1100), (np.float32, 2100)])
def test_path_trace_with_several_strips_and_chunks(dtype, n):
    """Sizes at which relax_k runs several column strips (the row-k snapshot is written by the
    first chunk of EVERY strip, the column-k snapshot by the first strip of every chunk).  The
    market-like input has no exact ties, so the reference's list is the next-hop walk: the trace
    must reproduce it for every sampled pair, and the product of the input rates along it must be
    the solved rate up to rounding."""
    rate, nxt, _ = synth.make("d2", n, dtype, seed=91)
    with engine.DeviceMatrix(n, dtype, with_next=True) as dm:
        dm.enable_path_log()
        dm.upload(rate, nxt)
        u = dm.solve(count_updates=True)
        assert u == dm.path_log_count() > 0
        r, nx, _ = dm.download()
        rnd = np.random.default_rng(n)
        # the batch form answers 20000 pairs in one launch: every list must end at its destination
        # and agree with the one-pair form
        bs, bd = rnd.integers(0, n, size=20000), rnd.integers(0, n, size=20000)
        batch = dm.query_exact_batch(bs, bd, cap=64)
        for q in range(0, 20000, 997):
            assert batch[q] == dm.query_exact(int(bs[q]), int(bd[q]))[1]
        assert all((len(p) == 0) == (int(bs[q]) == int(bd[q])) and (not p or p[-1] == int(bd[q]))
                   for q, p in enumerate(batch))
        longest = 0
        for _ in range(400):
            i, j = (int(x) for x in rnd.integers(0, n, size=2))
            q_rate, q_path = dm.query_exact(i, j)
            if i == j:
                assert q_path == []
                continue
            assert q_path == engine.follow_path(nx, i, j) and q_path[-1] == j
            prod, cur = 1.0, i
            for v in q_path:
                prod *= float(rate[cur, v])
                cur = v
            assert abs(prod - float(r[i, j])) <= 1e-5 * abs(float(r[i, j]))
            longest = max(longest, len(q_path))
        assert longest >= 3


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["d2", "t1", "t2", "t3"])
def test_hops_from_the_fused_engine(kind, dtype):
    """`hops` (= length _path) from the fused engine: the panel kernels carry the hops of the pivot
    rows / columns beside their rates and export their time-k snapshots; the main kernel forms
    hops = hops[i][k] + hops[k][j] at the winning pivot.  Same integers as the per-k engine forms
    step by step -- ties, unreachable pairs, arbitrage blow-ups (wrapping sums; t3 is outside the
    domain and runs on the per-k engine) -- through the host-buffer API (odd sizes padded) and the
    handle API."""
    for n in (300, 257):
        rate, nxt, hops = synth.make(kind, n, dtype, seed=400 + n)
        _solve_and_compare(rate, nxt, hops, engine=engine.FWX_ENGINE_FUSED)
    n = 384
    rate, nxt, hops = synth.make(kind, n, dtype, seed=9)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    oracle.relax(er, en, eh)
    with engine.DeviceMatrix(n, dtype, with_next=True, with_hops=True) as dm:
        dm.upload(rate, nxt, hops)
        dm.solve(engine=engine.FWX_ENGINE_FUSED)
        r, nx, hp = dm.download()
        assert_bits_equal(r, er, "rate")
        assert np.array_equal(nx, en) and np.array_equal(hp, eh)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hops_auto_large(dtype):
    """AUTO with hops at n = 3102 (not a multiple of 4: padded; full-size tiles): the fused engine
    must give the per-k engine's rate, next and hops (the per-k engine itself is checked against the
    oracle throughout this file)."""
    rate, nxt, hops = synth.make("d2", 3102, dtype, seed=77)
    a = [rate.copy(), nxt.copy(), hops.copy()]
    b = [rate.copy(), nxt.copy(), hops.copy()]
    ua = engine.solve(*a, count_updates=True)                                  # AUTO: fused route
    ub = engine.solve(*b, count_updates=True, engine=engine.FWX_ENGINE_PERK)
    assert ua == ub
    assert_bits_equal(a[0], b[0], "rate")
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert int(a[2].max()) >= 3
