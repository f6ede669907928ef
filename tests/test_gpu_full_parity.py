"""Whole-solve parity at the BASELINE.json configuration sizes: the ENTIRE solve runs on the oracle
(oracle.relax_mt: the C restatement of Algorithms.hs:42-61, rows of each pivot step split over the
host cores; here tiled over 16 pivots, oracle.relax_mt_tiled, which tests/test_oracle_golden.py pins to the
plain loop: several times faster, because the plain loop streams the matrix through host memory once per pivot) and every engine's result is compared with it bit for bit -- not pivot slices, not one
engine against another.  VERDICT r1 "close the parity chain at config sizes".

CPU cost on the GPU box's 16 cores: N=4096 ~4-8 s per dtype, N=8192 ~35 s.  The N=16384 solve
(~4 min of CPU) is run once by tools/full_parity_n16384.py; its record is profiles/r02_full_parity_n16384.json.
"""
import os

import numpy as np
import pytest

import oracle
from floydwarshall_amd import engine, synth

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


def _threads():
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_n4096_whole_solve_rate_next_hops_every_engine(dtype):
    """N = 4096, rates + next + hops, D2 (market-like: long winning paths): the whole oracle solve
    against PERK, FUSED (hops carried through the panels) and AUTO, all fields, and U."""
    n = 4096
    rate, nxt, hops = synth.make("d2", n, dtype, seed=synth.BASE_SEED + 41)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    eu = oracle.relax_mt_tiled(er, en, hops=eh, threads=_threads())
    for eng in (engine.FWX_ENGINE_PERK, engine.FWX_ENGINE_FUSED, engine.FWX_ENGINE_AUTO):
        gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
        u = engine.solve(gr, gn, gh, engine=eng, count_updates=True)
        assert_bits_equal(gr, er, "rate, engine %d" % eng)
        assert_bits_equal(gn, en, "next, engine %d" % eng)
        assert_bits_equal(gh, eh, "hops, engine %d" % eng)
        assert u == eu
    # rates only: the max-form kernels (f32 v_max3 pairs, f64 v_max_f64)
    gr = rate.copy()
    engine.solve(gr, engine=engine.FWX_ENGINE_FUSED)
    assert_bits_equal(gr, er, "rate, fused max form")


def test_config3_n8192_fp32_whole_solve_vs_oracle():
    """BASELINE config 3 (N = 8192 fp32): whole oracle solve against the fused engine (rates only =
    max form; rates + next = compare form) and the per-k engine, bit for bit."""
    n = 8192
    rate, nxt, _ = synth.make("d1", n, np.float32, seed=synth.BASE_SEED + 2)
    er, en = rate.copy(), nxt.copy()
    eu = oracle.relax_mt_tiled(er, en, threads=_threads())
    gr = rate.copy()
    engine.solve(gr, engine=engine.FWX_ENGINE_FUSED)
    assert_bits_equal(gr, er, "fused rates-only")
    gr, gn = rate.copy(), nxt.copy()
    u = engine.solve(gr, gn, engine=engine.FWX_ENGINE_FUSED, count_updates=True)
    assert_bits_equal(gr, er, "fused rate")
    assert_bits_equal(gn, en, "fused next")
    assert u == eu
    gr, gn = rate.copy(), nxt.copy()
    u = engine.solve(gr, gn, engine=engine.FWX_ENGINE_PERK, count_updates=True)
    assert_bits_equal(gr, er, "per-k rate")
    assert_bits_equal(gn, en, "per-k next")
    assert u == eu


def test_n8192_f64_whole_solve_rate_next_hops_perk_and_fused():
    """The reference's own precision (Types.hs:26) at config 3's order: N = 8192 f64, rates + next + hops,
    the WHOLE oracle solve (~1 min on the box's host cores) against the per-k engine and the fused engine
    (counted: compare form; uncounted: the f64 arg kernels and, rates only, the double-pass max form)."""
    n = 8192
    rate, nxt, hops = synth.make("d1", n, np.float64, seed=synth.BASE_SEED + 43)
    er, en, eh = rate.copy(), nxt.copy(), hops.copy()
    eu = oracle.relax_mt_tiled(er, en, hops=eh, threads=_threads())
    for eng in (engine.FWX_ENGINE_PERK, engine.FWX_ENGINE_FUSED):
        gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
        u = engine.solve(gr, gn, gh, engine=eng, count_updates=True)
        assert_bits_equal(gr, er, "rate, engine %d" % eng)
        assert_bits_equal(gn, en, "next, engine %d" % eng)
        assert_bits_equal(gh, eh, "hops, engine %d" % eng)
        assert u == eu
    gr, gn, gh = rate.copy(), nxt.copy(), hops.copy()
    engine.solve(gr, gn, gh, engine=engine.FWX_ENGINE_FUSED)              # fused_main_arg_f64 + hops
    assert_bits_equal(gr, er, "rate, fused arg form")
    assert_bits_equal(gn, en, "next, fused arg form")
    assert_bits_equal(gh, eh, "hops, fused arg form")
    gr = rate.copy()
    engine.solve(gr)                                                       # AUTO, rates only: double-pass max form
    assert_bits_equal(gr, er, "rate, fused max form (double pass)")


def test_config4_matrix_in_f64_against_the_whole_oracle_digests():
    """The headline matrix at the reference's precision: N = 16384 f64 (bench.py's `f64` leg), rates +
    next-hops, against the committed digests of ONE WHOLE f64 oracle solve
    (tests/golden/config4_n16384_f64_digests.json, made by make_config4_digests.py --f64 --next on the GPU
    box's host cores): fused engine with next-hops, fused rates-only (double pass), per-k engine."""
    from helpers import digest, load_golden
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_n16384_f64_digests.json")
    if not os.path.exists(path):
        pytest.skip("fixture not generated yet")
    gold = load_golden("config4_n16384_f64_digests.json")
    n = gold["n"]
    rate, nxt = synth.d1_uniform(n, np.float64, synth.BASE_SEED + 3)
    gr, gn = rate.copy(), nxt.copy()
    engine.solve(gr, gn, engine=engine.FWX_ENGINE_FUSED)
    assert digest(gr) == gold["rate_digest"] and digest(gn) == gold["next_digest"]
    gr = rate.copy()
    engine.solve(gr)
    assert digest(gr) == gold["rate_digest"]
    gr, gn = rate.copy(), nxt.copy()
    u = engine.solve(gr, gn, engine=engine.FWX_ENGINE_PERK, count_updates=True)
    assert digest(gr) == gold["rate_digest"] and digest(gn) == gold["next_digest"] and u == gold["U"]
