"""CPU ORACLE for the max-product Floyd-Warshall hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package.
The product (floydwarshall_amd/) never imports, links or calls anything here, and has no CPU
fallback: it fails loudly when the HIP library is missing.

Contents
  fw_oracle.c / libfworacle.so   dense in-place restatement of runAlgo
                                 (/root/reference/src/lib/Algorithms.hs:42-61), f64 and f32,
                                 single- and multi-threaded (the latter also with a chunk pre-check
                                 for the big continuation tests); copy-per-k literal form.
  list_faithful.py               entry-for-entry restatement with whole `_path` lists of
                                 buildMatrix / runAlgo / floydWarshall / optimum
                                 (Algorithms.hs:19-78).

Parity pinning: the reference is Haskell and cannot be built here (no GHC/cabal/nix in the
image), so the oracle is pinned by the reference's own golden vectors committed as data under
tests/golden/ (tests/test_oracle_golden.py), and above N=4 by agreement of the two independent
restatements (dense C vs list-faithful Python).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile libfworacle.so with oracle/Makefile (gcc only)."""
    subprocess.run(["make", "-s", "-C", _HERE, "libfworacle.so"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfworacle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i32, u64, vp = ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p
        for name in ("fwo_relax_f64", "fwo_relax_f32"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [i32, vp, vp, vp, i32, i32]
        for name in ("fwo_relax_mt_f64", "fwo_relax_mt_f32", "fwo_relax_mt_fast_f64", "fwo_relax_mt_fast_f32"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [i32, vp, vp, vp, i32, i32, i32]
        for name in ("fwo_relax_mt_tiled_f64", "fwo_relax_mt_tiled_f32"):
            getattr(L, name).restype = ctypes.c_int64
            getattr(L, name).argtypes = [i32, vp, vp, vp, i32, i32, i32, i32]
        for name in ("fwo_copy_per_k_f64", "fwo_copy_per_k_f32"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [i32, vp, vp, vp]
        L.fwo_follow_path.restype = i32
        L.fwo_follow_path.argtypes = [i32, vp, i32, i32, vp, i32]
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _check(rate, nxt, hops):
    assert rate.ndim == 2 and rate.shape[0] == rate.shape[1]
    assert rate.dtype in (np.float32, np.float64) and rate.flags.c_contiguous
    for a in (nxt, hops):
        if a is not None:
            assert a.shape == rate.shape and a.dtype == np.int32 and a.flags.c_contiguous
    return "f64" if rate.dtype == np.float64 else "f32"


def relax(rate, nxt=None, hops=None, k_begin=0, k_end=None):
    """In-place dense k-i-j over pivots [k_begin, k_end); returns U (successful relaxations)."""
    sfx = _check(rate, nxt, hops)
    n = rate.shape[0]
    k_end = n if k_end is None else k_end
    return int(getattr(lib(), "fwo_relax_" + sfx)(n, _ptr(rate), _ptr(nxt), _ptr(hops),
                                                  k_begin, k_end))


def relax_mt(rate, nxt=None, k_begin=0, k_end=None, threads=None, hops=None, fast=False):
    """Same result as relax(), rows of each pivot step split over `threads` host threads.  fast: the same loop
    with a side-effect-free pre-check that skips 64-column chunks in which the compare of Algorithms.hs:55 holds
    for no entry (fwo_relax_mt_fast_*; pinned to the plain loop by tests/test_oracle_golden.py) -- for the tests
    that continue solves at N = 8192 ... 32768."""
    sfx = _check(rate, nxt, hops)
    assert hops is None or nxt is not None
    n = rate.shape[0]
    k_end = n if k_end is None else k_end
    threads = threads or len(os.sched_getaffinity(0))
    name = ("fwo_relax_mt_fast_" if fast else "fwo_relax_mt_") + sfx
    return int(getattr(lib(), name)(n, _ptr(rate), _ptr(nxt), _ptr(hops), k_begin, k_end, threads))


def relax_mt_tiled(rate, nxt=None, k_begin=0, k_end=None, threads=None, hops=None, tile=16):
    """Same result and U as relax(): the multi-threaded loop tiled over `tile` pivots (each row takes the pivots
    of a tile in one visit, from snapshots of the pivot rows taken at their own time; fwo_relax_mt_tiled_*,
    pinned to the plain loop by tests/test_oracle_golden.py) -- for the stretches of a solve at N = 32768, where
    every pivot step of the plain loop streams the whole matrix through host memory."""
    sfx = _check(rate, nxt, hops)
    assert hops is None or nxt is not None
    n = rate.shape[0]
    k_end = n if k_end is None else k_end
    threads = threads or len(os.sched_getaffinity(0))
    u = int(getattr(lib(), "fwo_relax_mt_tiled_" + sfx)(n, _ptr(rate), _ptr(nxt), _ptr(hops), k_begin, k_end,
                                                        threads, int(tile)))
    if u < 0:
        raise MemoryError("fwo_relax_mt_tiled: snapshot buffers")
    return u


def copy_per_k(rate, nxt, hops):
    """Literal new-matrix-per-k form (Algorithms.hs:44); small n."""
    sfx = _check(rate, nxt, hops)
    rc = getattr(lib(), "fwo_copy_per_k_" + sfx)(rate.shape[0], _ptr(rate), _ptr(nxt), _ptr(hops))
    assert rc == 0


def follow_path(nxt, src, dst):
    """Index path src -> dst by following next-hops; [] if unreachable; None on a cycle."""
    n = nxt.shape[0]
    out = np.empty(max(n, 1), dtype=np.int32)
    ln = lib().fwo_follow_path(n, _ptr(nxt), int(src), int(dst), _ptr(out), n)
    if ln < 0:
        return None
    return [int(x) for x in out[:ln]]
