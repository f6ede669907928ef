"""Pins the ORACLE (CPU restatement) against the reference's own golden vectors, and the two
independent restatements (dense C, list-faithful Python) against each other.  CPU only.

Golden data: tests/golden/*.json, transcribed (inputs + expected outputs only) from
/root/reference/src/test/AlgorithmsTest.hs and MockData.hs -- see each file's `_source`.
"""
import random

import numpy as np
import pytest

import oracle
from floydwarshall_amd import synth
from oracle import list_faithful as lf

from helpers import (assert_bits_equal, golden_dense, golden_rates_dict, load_golden)


def test_build_matrix_4x4_matches_reference_golden():
    # AlgorithmsTest.hs:49-60 (buildMatrix_4x4Matrix)
    g = load_golden("algorithms_4x4.json")
    m = lf.build_matrix(golden_rates_dict(g))
    vertices, rate, nxt, hops = lf.to_dense(m)
    assert [list(v) for v in vertices] == g["vertices"]
    erate, enext, ehops, epaths = golden_dense(g["initial"])
    assert_bits_equal(rate, erate, "initial rate")
    assert np.array_equal(nxt, enext) and np.array_equal(hops, ehops)
    assert lf.path_indices(m) == epaths


def test_floyd_warshall_4x4_matches_reference_golden_list_faithful():
    # AlgorithmsTest.hs:66-77 (floydWarshall_4x4Matrix)
    g = load_golden("algorithms_4x4.json")
    m = lf.floyd_warshall(golden_rates_dict(g))
    _, rate, nxt, hops = lf.to_dense(m)
    erate, enext, ehops, epaths = golden_dense(g["solved"])
    assert_bits_equal(rate, erate, "solved rate")
    assert lf.path_indices(m) == epaths
    assert np.array_equal(nxt, enext) and np.array_equal(hops, ehops)


@pytest.mark.parametrize("form", ["inplace", "copy_per_k", "mt"])
def test_floyd_warshall_4x4_matches_reference_golden_dense_c(form):
    g = load_golden("algorithms_4x4.json")
    rate, nxt, hops, _ = golden_dense(g["initial"])
    if form == "inplace":
        u = oracle.relax(rate, nxt, hops)
        assert u == 8
    elif form == "copy_per_k":
        oracle.copy_per_k(rate, nxt, hops)
    else:
        oracle.relax_mt(rate, nxt, threads=3)
        hops = None
    erate, enext, ehops, epaths = golden_dense(g["solved"])
    assert_bits_equal(rate, erate, "solved rate")
    assert np.array_equal(nxt, enext)
    if hops is not None:
        assert np.array_equal(hops, ehops)
    # following next-hops reproduces every golden `_path` list
    for i in range(4):
        for j in range(4):
            assert tuple(oracle.follow_path(nxt, i, j)) == epaths[i][j]


def test_empty_map_gives_empty_matrix():
    # AlgorithmsTest.hs:45-47, :62-64
    assert lf.build_matrix({}) == []
    assert lf.floyd_warshall({}) == []
    rate = np.zeros((0, 0))
    assert oracle.relax(rate) == 0


def test_optimum_golden_cases():
    g = load_golden("algorithms_4x4.json")
    c = load_golden("optimum_cases.json")
    m = lf.floyd_warshall(golden_rates_dict(g))
    for case in c["not_exist"]:                                  # AlgorithmsTest.hs:82-91
        assert lf.optimum(tuple(case["src"]), tuple(case["dst"]), m) == ("err", case["err"])
    i, j = c["reachability"]["isolate"]                          # AlgorithmsTest.hs:93-110
    m2 = [list(r) for r in m]
    m2[i][j] = (0.0, m2[i][j][1], ())
    for case in c["reachability"]["cases"]:
        res = lf.optimum(tuple(case["src"]), tuple(case["dst"]), m2)
        if "err" in case:
            assert res == ("err", case["err"])
        else:
            assert res[0] == "ok"
            assert res[1][0] == case["rate"]
            assert [list(v) for v in res[1][2]] == case["path"]


def test_optimum_error_precedence_property():
    # AlgorithmsTest.hs:112-134 with the generators of MockData.hs:59-81, restated
    c = load_golden("optimum_cases.json")
    sample = [tuple(v) for v in c["sample_vertices"]]
    rnd = random.Random(7)
    for _ in range(400):
        src, dest = rnd.choice(sample), rnd.choice(sample)
        k = rnd.randint(0, len(sample) // 2 + 1)
        vertices = sorted(set(rnd.choice(sample) for _ in range(k)))
        if rnd.random() < 0.5:
            matrix = [[] for _ in vertices]
        else:
            matrix = [[(0.0, s, ()) if s == d else (1.0, s, (d,)) for d in vertices]
                      for s in vertices]
        res = lf.optimum(src, dest, matrix)
        sv, dv = lf.show_vertex(src), lf.show_vertex(dest)
        if len(matrix) == 0:
            assert res == ("err", sv + " is not entered before")
        elif any(len(r) == 0 for r in matrix):
            assert res == ("err", "The matrix is empty")
        elif src not in vertices:
            assert res == ("err", sv + " is not entered before")
        elif dest not in vertices:
            assert res == ("err", dv + " is not entered before")
        elif src == dest:
            assert res == ("err", "There is no exchange between " + sv + " and " + dv)
        else:
            assert res == ("ok", (1.0, src, (dest,)))


@pytest.mark.parametrize("kind", ["d1", "d2", "t1", "t2", "t3"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_c_equals_list_faithful(kind, dtype):
    """Above N=4 the reference pins nothing: the two restatements must agree bit for bit, in
    rates, in head-of-path (next) and in path length (hops)."""
    n = 24
    rate, nxt, hops = synth.make(kind, n, dtype, seed=100 + n)
    vertices = [("X", "C%03d" % i) for i in range(n)]
    m = lf.run_algo(lf.from_dense(vertices, rate, nxt), dtype)
    _, lrate, lnext, lhops = lf.to_dense(m, dtype)
    r2, n2, h2 = rate.copy(), nxt.copy(), hops.copy()
    oracle.relax(r2, n2, h2)
    assert_bits_equal(r2, lrate, "rate %s" % kind)
    assert np.array_equal(n2, lnext)
    assert np.array_equal(h2, lhops)
    if kind in ("d1", "d2", "t1", "t2"):
        # no arbitrage: following the final next-hops reproduces every whole `_path` list
        paths = lf.path_indices(m)
        for i in range(n):
            for j in range(n):
                assert tuple(oracle.follow_path(n2, i, j)) == paths[i][j]


def test_empty_ikpath_takes_the_head_of_kjpath():
    """`head (ikPath ++ kjPath)` (Algorithms.hs:55) is head kjPath when ikPath is empty.
    rate[0][1] = -0.75 with path [1], rate[0][2] = 0 with the empty path, rate[2][1] = 2 with path
    [1]: step 2 improves (0,1) to 0*2 = 0 > -0.75 and the list becomes [] ++ [1] = [1]."""
    rate = np.array([[0, -0.75, 0], [0, 0, 0], [0, 2, 0]], dtype=np.float64)
    nxt = np.array([[-1, 1, -1], [-1, -1, -1], [-1, 1, -1]], dtype=np.int32)
    hops = (nxt >= 0).astype(np.int32)
    vertices = [("X", "C%d" % i) for i in range(3)]
    m = lf.run_algo(lf.from_dense(vertices, rate, nxt), np.float64)
    _, lrate, lnext, lhops = lf.to_dense(m, np.float64)
    assert lrate[0, 1] == 0.0 and lnext[0, 1] == 1 and lhops[0, 1] == 1
    for solve in (oracle.relax, oracle.copy_per_k):
        r2, n2, h2 = rate.copy(), nxt.copy(), hops.copy()
        solve(r2, n2, h2)
        assert_bits_equal(r2, lrate, "rate")
        assert np.array_equal(n2, lnext) and np.array_equal(h2, lhops)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_c_equals_list_faithful_on_hostile_values(dtype):
    """Outside the reference's domain (negative rates, NaN, inf, -0, a positive rate whose path is
    empty) the dense oracle must still be the reference's loop: compare it with the list-faithful
    restatement, which concatenates whole lists and knows nothing of next-hops."""
    from hostile_inputs import hostile_matrix
    rnd = np.random.default_rng(99)
    with np.errstate(all="ignore"):
        for n in (3, 5, 8, 13, 21):
            for _ in range(12):
                rate, nxt, hops = hostile_matrix(rnd, n, dtype)
                vertices = [("X", "C%03d" % i) for i in range(n)]
                m = lf.run_algo(lf.from_dense(vertices, rate, nxt), dtype)
                _, lrate, lnext, lhops = lf.to_dense(m, dtype)
                r2, n2, h2 = rate.copy(), nxt.copy(), hops.copy()
                oracle.relax(r2, n2, h2)
                assert_bits_equal(r2, lrate, "rate")
                assert np.array_equal(n2, lnext), (n2, lnext)
                assert np.array_equal(h2, lhops)
                r3, n3, h3 = rate.copy(), nxt.copy(), hops.copy()
                oracle.relax_mt(r3, n3, hops=h3, threads=3)
                assert_bits_equal(r3, lrate, "mt rate")
                assert np.array_equal(n3, lnext) and np.array_equal(h3, lhops)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_inplace_equals_copy_per_k_and_mt(dtype):
    for kind in ("d1", "t1", "t3"):
        rate, nxt, hops = synth.make(kind, 61, dtype, seed=5)
        a = (rate.copy(), nxt.copy(), hops.copy())
        b = (rate.copy(), nxt.copy(), hops.copy())
        c = (rate.copy(), nxt.copy())
        ua = oracle.relax(*a)
        oracle.copy_per_k(*b)
        uc = oracle.relax_mt(*c, threads=4)
        assert ua == uc
        assert_bits_equal(a[0], b[0], "copy_per_k rate")
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert_bits_equal(a[0], c[0], "mt rate")
        assert np.array_equal(a[1], c[1])


def test_k_range_is_resumable():
    rate, nxt, hops = synth.make("d2", 50, np.float64, seed=9)
    a = (rate.copy(), nxt.copy(), hops.copy())
    b = (rate.copy(), nxt.copy(), hops.copy())
    oracle.relax(*a)
    oracle.relax(*b, k_begin=0, k_end=17)
    oracle.relax(*b, k_begin=17, k_end=50)
    assert_bits_equal(a[0], b[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_the_chunk_pre_check_loop_equals_the_plain_loop(dtype):
    """oracle/fw_oracle_fast.c (relax_mt(..., fast=True) and relax_mt_tiled: the loops the big continuation tests
    use) against the plain restatement fwo_relax_*: every input kind -- ordinary, market-like, ties, sparse, arbitrage / inf / NaN,
    overflow --, sizes around the 64-column chunk, with and without next / hops, pivot ranges, thread counts;
    rates, next, hops and U, bit for bit.  And the reference's own golden 4 x 4."""
    for kind in ("d1", "d2", "t1", "t2", "t3", "t4"):
        for n in (1, 2, 63, 64, 65, 130, 257):
            rate, nxt, hops = synth.make(kind, n, dtype, seed=700 + n)
            for fields in (0, 1, 2):
                a = [rate.copy(), nxt.copy() if fields >= 1 else None, hops.copy() if fields >= 2 else None]
                b = [rate.copy(), nxt.copy() if fields >= 1 else None, hops.copy() if fields >= 2 else None]
                c = [rate.copy(), nxt.copy() if fields >= 1 else None, hops.copy() if fields >= 2 else None]
                ua = oracle.relax(*a)
                ub = oracle.relax_mt(b[0], b[1], hops=b[2], threads=1 + (n + fields) % 5, fast=True)
                # ... and the loop tiled over pivots (oracle.relax_mt_tiled: ragged tiles, tile 1 = the plain order)
                uc = oracle.relax_mt_tiled(c[0], c[1], hops=c[2], threads=1 + (n + 2 * fields) % 4,
                                           tile=(1, 5, 16, 64)[(n + fields) % 4])
                assert ua == ub == uc, (kind, n, fields)
                assert_bits_equal(a[0], b[0], "rate %s n=%d" % (kind, n))
                assert_bits_equal(a[0], c[0], "tiled rate %s n=%d" % (kind, n))
                if fields >= 1:
                    assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])
                if fields >= 2:
                    assert np.array_equal(a[2], b[2]) and np.array_equal(a[2], c[2])
    rate, nxt, hops = synth.make("d2", 200, dtype, seed=77)
    a, b = (rate.copy(), nxt.copy(), hops.copy()), (rate.copy(), nxt.copy(), hops.copy())
    ua = oracle.relax(*a, k_begin=37, k_end=150)
    ub = oracle.relax_mt(b[0], b[1], 37, 90, threads=3, hops=b[2], fast=True)
    ub += oracle.relax_mt(b[0], b[1], 90, 150, threads=7, hops=b[2], fast=True)
    assert ua == ub
    assert_bits_equal(a[0], b[0], "ranged rate")
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    c = (rate.copy(), nxt.copy(), hops.copy())
    uc = oracle.relax_mt_tiled(c[0], c[1], 37, 101, threads=4, hops=c[2], tile=16)     # ragged: 64 = 4 tiles, then
    uc += oracle.relax_mt_tiled(c[0], c[1], 101, 150, threads=2, hops=c[2], tile=20)   # 49 = 20 + 20 + 9
    assert ua == uc
    assert_bits_equal(a[0], c[0], "ranged tiled rate")
    assert np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
    from hostile_inputs import hostile_matrix
    rnd = np.random.default_rng(199)
    for case in range(60):
        n = int(rnd.integers(2, 91))
        r, nx, hp = hostile_matrix(rnd, n, dtype)
        a, b = (r.copy(), nx.copy(), hp.copy()), (r.copy(), nx.copy(), hp.copy())
        with np.errstate(all="ignore"):
            ua = oracle.relax(*a)
            ub = oracle.relax_mt(b[0], b[1], hops=b[2], threads=2, fast=True)
            c = (r.copy(), nx.copy(), hp.copy())
            uc = oracle.relax_mt_tiled(c[0], c[1], hops=c[2], threads=3, tile=1 + case % 9)
        assert ua == ub == uc, case
        assert_bits_equal(a[0], b[0], "hostile rate, case %d" % case)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert_bits_equal(a[0], c[0], "hostile tiled rate, case %d" % case)
        assert np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
    g = load_golden("algorithms_4x4.json")
    rate, nxt, hops, _ = golden_dense(g["initial"], dtype)
    oracle.relax_mt(rate, nxt, hops=hops, threads=2, fast=True)
    erate, enext, ehops, _ = golden_dense(g["solved"], dtype)
    if dtype == np.float64:
        assert_bits_equal(rate, erate, "golden solved 4x4")
    assert np.array_equal(nxt, enext) and np.array_equal(hops, ehops)
