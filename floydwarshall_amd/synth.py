"""Seeded synthetic rate matrices for the benchmark and the parity tests.

All generators draw in float64 from numpy's PCG64 with an explicit seed and then round to the
working dtype, so host, oracle and device see identical bits.  The matrices have the shape
buildMatrix (/root/reference/src/lib/Algorithms.hs:26-40) produces: diagonal 0.0 / next -1,
off-diagonal rate / next j (or 0.0 / -1 where there is no edge).

  d1_uniform     rate ~ U(0.05, 1.0]: every cycle product <= 1 (no arbitrage, no overflow)
  d2_market      rate[i][j] = p[j]/p[i] * U(0.90, 1.0], p ~ U(0.5, 2): FX quotes with a spread
  t1_ties        rate = 2^-e, e in {0..3}: exact ties everywhere (earliest k must win)
  t2_sparse_ties t1 at 15 % density, zeros = unreachable
  t3_arbitrage   rates up to 4 with inf and NaN sprinkled in: overflow / NaN semantics
  t4_overflow    non-negative NaN-free rates up to e^60 at 70 % density: +inf and inf*0 = NaN arise
"""
import numpy as np

BASE_SEED = 20240  # seed = BASE_SEED + BASELINE.json config index


def _finish(rate64, dtype, present=None):
    n = rate64.shape[0]
    rate = rate64.astype(dtype)
    nxt = np.tile(np.arange(n, dtype=np.int32), (n, 1))
    if present is not None:
        rate[~present] = 0.0
        nxt[~present] = -1
    np.fill_diagonal(rate, 0.0)
    np.fill_diagonal(nxt, -1)
    return np.ascontiguousarray(rate), np.ascontiguousarray(nxt)


def _uniform(rng, shape, lo, hi, chunk_rows=2048):
    """(lo, hi] uniform, generated in row chunks to bound peak memory at large n."""
    out = np.empty(shape, dtype=np.float64)
    for r in range(0, shape[0], chunk_rows):
        blk = rng.random((min(chunk_rows, shape[0] - r),) + tuple(shape[1:]))
        out[r:r + blk.shape[0]] = hi - blk * (hi - lo)
    return out


def d1_uniform(n, dtype=np.float32, seed=BASE_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    return _finish(_uniform(rng, (n, n), 0.05, 1.0), dtype)


def d2_market(n, dtype=np.float32, seed=BASE_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    p = 0.5 + 1.5 * rng.random(n)
    spread = _uniform(rng, (n, n), 0.90, 1.0)
    return _finish(spread * (p[None, :] / p[:, None]), dtype)


def t1_ties(n, dtype=np.float32, seed=BASE_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    return _finish(np.ldexp(1.0, -rng.integers(0, 4, size=(n, n))), dtype)


def t2_sparse_ties(n, dtype=np.float32, seed=BASE_SEED, density=0.15):
    rng = np.random.Generator(np.random.PCG64(seed))
    rate = np.ldexp(1.0, -rng.integers(0, 4, size=(n, n)))
    present = rng.random((n, n)) < density
    return _finish(rate, dtype, present)


def t3_arbitrage(n, dtype=np.float32, seed=BASE_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    rate = 4.0 * rng.random((n, n))
    special = rng.random((n, n))
    rate[special < 0.01] = np.inf
    rate[(special >= 0.01) & (special < 0.02)] = np.nan
    rate[(special >= 0.02) & (special < 0.03)] = -1.5
    return _finish(rate, dtype)


def t4_overflow(n, dtype=np.float32, seed=BASE_SEED):
    """Non-negative, NaN-free, but huge: products overflow to +inf and inf * 0 = NaN candidates
    appear.  Inside the max-form kernel's domain, at its edge."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rate = np.exp(rng.uniform(-5.0, 60.0, size=(n, n)))
    present = rng.random((n, n)) < 0.7
    return _finish(rate, dtype, present)


GENERATORS = {"d1": d1_uniform, "d2": d2_market, "t1": t1_ties, "t2": t2_sparse_ties,
              "t3": t3_arbitrage, "t4": t4_overflow}


def make(kind, n, dtype=np.float32, seed=BASE_SEED):
    """(rate, next, hops) for one of the named distributions; hops = 1 where an edge exists."""
    rate, nxt = GENERATORS[kind](n, dtype, seed)
    hops = (nxt >= 0).astype(np.int32)
    return rate, nxt, hops
