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


# ---- device memory for the tests: the HIP runtime through ctypes, no torch in the process ----------
def dev(a):
    """numpy array -> device array (floydwarshall_amd.hip.DeviceArray)."""
    from floydwarshall_amd import hip
    return hip.DeviceArray.from_numpy(np.ascontiguousarray(a))


def dev_empty(shape, dtype):
    from floydwarshall_amd import hip
    return hip.DeviceArray(shape, dtype)


def dev_zeros(shape, dtype):
    from floydwarshall_amd import hip
    return hip.DeviceArray(shape, dtype).zero_(hip.default_stream())


def host(d):
    """device array -> numpy (waits for the default stream first)."""
    return d.numpy()


def host_cat(ds):
    return np.concatenate([d.numpy() for d in ds])


def dev_sync():
    from floydwarshall_amd import hip
    hip.synchronize()


def digest(a):
    """xxh64 of an array's bytes, in 64 MiB chunks ("xxh64:<hex>": the form the committed fixtures use)."""
    import xxhash
    buf = memoryview(np.ascontiguousarray(a)).cast("B")
    h = xxhash.xxh64()
    for off in range(0, len(buf), 1 << 26):
        h.update(buf[off:off + (1 << 26)])
    return "xxh64:" + h.hexdigest()


def spawn_ranks(fn, args, nprocs, timeout=600):
    """Start `nprocs` fresh interpreters ("spawn": nothing of this process's GPU state is inherited)
    running fn(rank, *args); every rank must exit with 0.  The stdlib twin of
    torch.multiprocessing.spawn -- the pytest process itself never imports torch, so libfwx stays on
    the HIP runtime it was built against here while the ranks are free to import torch first."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=fn, args=(rank,) + tuple(args)) for rank in range(nprocs)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    bad = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode != 0]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not bad, "ranks failed (rank, exit code): %s" % bad


def path_from_trace(last, at_col, at_row, next0, a, b):
    """Host restatement of exact_path_kernel: path(a,b) = path_q(a,q) ++ path_q(q,b), q = newest
    pivot of (a,b) -- `last` for the query, at_col / at_row for the two halves."""
    out = []
    stack = [(a, b, 0)]
    while stack:
        x, y, kind = stack.pop()
        q = (last, at_col, at_row)[kind][x, y]
        if q < 0:
            if next0[x, y] >= 0:
                out.append(y)
        else:
            stack.append((int(q), y, 2))
            stack.append((x, int(q), 1))
    return out
