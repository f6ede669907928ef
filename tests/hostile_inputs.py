"""Input generators shared by the CPU and GPU parity tests (no test functions here)."""
import numpy as np


def hostile_matrix(rnd, n, dtype):
    """Entries drawn from a pool of awkward values: zeros of both signs, ties, subnormals, values
    that overflow when multiplied, infinities, NaN, negatives.  The diagonal is arbitrary too
    (the reference never reads or writes it, Algorithms.hs:54)."""
    fi = np.finfo(dtype)
    pool = np.array([0.0, -0.0, 1.0, 1.0, 0.5, 0.5, 2.0, 0.25, 3.0, 1e-3, 7.0, fi.tiny, fi.tiny / 4,
                     fi.max / 2, fi.max, np.inf, -np.inf, np.nan, -1.0, -0.5, 1.5, 0.999], dtype=dtype)
    heavy = rnd.random() < 0.5          # half of the cases: mostly ordinary rates, a few oddities
    p = np.ones(len(pool))
    if heavy:
        p[2:11] = 12.0
    rate = rnd.choice(pool, size=(n, n), p=p / p.sum()).astype(dtype)
    nxt = np.where(rnd.random((n, n)) < 0.8, np.arange(n, dtype=np.int32)[None, :], -1).astype(np.int32)
    hops = (nxt >= 0).astype(np.int32)
    return np.ascontiguousarray(rate), np.ascontiguousarray(nxt), np.ascontiguousarray(hops)
