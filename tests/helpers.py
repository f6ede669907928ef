"""Shared helpers for the parity tests (the oracle is the checker, never the thing tested)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def golden_rates_dict(g):
    return {(tuple(r["src"]), tuple(r["dst"])): r["rate"] for r in g["rates"]}


def golden_dense(entries, dtype=np.float64):
    """[[ [rate,[path idx]] ]] -> rate, next (= head of path or -1), hops (= len path), paths."""
    n = len(entries)
    rate = np.zeros((n, n), dtype=dtype)
    nxt = np.full((n, n), -1, dtype=np.int32)
    hops = np.zeros((n, n), dtype=np.int32)
    paths = [[tuple(e[1]) for e in row] for row in entries]
    for i in range(n):
        for j in range(n):
            rate[i, j] = entries[i][j][0]
            p = entries[i][j][1]
            hops[i, j] = len(p)
            if p:
                nxt[i, j] = p[0]
    return rate, nxt, hops, paths


def bits_equal(a, b):
    """Bit-exact comparison that treats NaNs by payload, and -0.0 != +0.0."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind == "f":
        it = np.uint64 if a.dtype.itemsize == 8 else np.uint32
        return bool(np.array_equal(a.view(it), b.view(it)))
    return bool(np.array_equal(a, b))


def assert_bits_equal(a, b, what=""):
    if not bits_equal(a, b):
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        if a.dtype.kind == "f":
            it = np.uint64 if a.dtype.itemsize == 8 else np.uint32
            diff = a.view(it) != b.view(it)
        else:
            diff = a != b
        idx = np.argwhere(diff)
        first = tuple(idx[0]) if len(idx) else None
        raise AssertionError("%s: %d entries differ; first at %s: %r vs %r" % (
            what, int(diff.sum()), first, a[first] if first else None, b[first] if first else None))
