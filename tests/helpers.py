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


# ---- BASELINE config 5 (N = 32768 f32, rate + next + hops): oracle slices around launch boundaries -----
CONFIG5_SLICES = ((0, 256), (16256, 16512), (32512, 32768))


def config5_solve_with_oracle_slices(dm, n, threads=None):
    """Solves the matrix uploaded into the handle `dm` (rate + next + hops) over all n pivots, and pins
    three 256-pivot stretches to the oracle: the first pivots (from the input itself), a stretch across the
    middle of the matrix (pivot 16384 = a partition boundary at P = 8; the stretch starts 128 before it, so
    a 128-pivot launch ends exactly there) and the last 256 pivots (ending at the solved matrix).  Every
    stretch is a pivot range of its own on the GPU -- four full blocks: two 128-pivot launches of the double
    pass, panels across a launch boundary -- and is continued on the oracle (rate + next + hops; the loop tiled
    over 16 pivots, oracle.relax_mt_tiled, which tests/test_oracle_golden.py pins to the plain loop: the plain
    loop streams 12 GiB through host memory per pivot) from the state the GPU held before it.  Returns the solved (rate, next, hops).

    Two tests (the plain handle, P = 8 partitions) walk the same input through the same stretches.  The
    first to run does the above and records the digests of the oracle's state after each stretch; the
    second only has to reach THOSE states: equal digests after a stretch mean its matrix equals one the
    oracle produced from a verified predecessor -- no second oracle run, no download before the stretch.
    The oracle runs on a thread of its own (ctypes releases the GIL) while the GPU solves the stretch and the
    result comes back over PCIe."""
    import threading
    import oracle
    pos = 0
    for a, b in CONFIG5_SLICES:
        if a > pos:
            dm.solve(k_begin=pos, k_end=a)
        cached = _CONFIG5_ORACLE.get((n, a, b))
        if cached is None:
            er, en, eh = dm.download()
            t = threading.Thread(target=oracle.relax_mt_tiled, args=(er, en, a, b),
                                 kwargs={"threads": threads, "hops": eh, "tile": 16})
            t.start()                                # the oracle continues from the GPU state, beside the GPU
        dm.solve(k_begin=a, k_end=b)
        gr, gn, gh = dm.download()
        if cached is None:
            t.join()
            assert_bits_equal(gr, er, "rate after pivots [%d, %d)" % (a, b))
            assert_bits_equal(gn, en, "next after pivots [%d, %d)" % (a, b))
            assert_bits_equal(gh, eh, "hops after pivots [%d, %d)" % (a, b))
            _CONFIG5_ORACLE[(n, a, b)] = (digest(er), digest(en), digest(eh))
            del er, en, eh
        else:
            assert (digest(gr), digest(gn), digest(gh)) == cached, \
                "state after pivots [%d, %d) differs from the oracle's continuation" % (a, b)
        pos = b
    assert pos == n
    return gr, gn, gh


_CONFIG5_ORACLE = {}     # (n, a, b) -> digests (rate, next, hops) of the oracle's state after the stretch


def check_walks_and_exact_lists(rate0, rate, nxt, hops, src, dst, walk_len, walk_prod, exact_lists_of):
    """What path reconstruction must satisfy at config 5's size, stated exactly:
    (1) every walk of the FINAL next-hops from src ends at dst (length >= 1 iff src != dst: D1 is dense);
    (2) its product of INPUT edge rates is the solved rate up to the fp32 roundings of the two routes
        involved: |walk product - rate| <= (len + hops) * 2^-24 * rate  (each of the multiplications that
        formed the stored rate, and each the walk's own route would take, rounds by at most 2^-24);
    (3) `hops` is the length of the list the reference concatenated when the entry last improved
        (Algorithms.hs:55).  The walk may be a different route of equal rank (sub-routes re-routed later),
        so len(walk) == hops is NOT a property -- but for EVERY sampled pair where they differ (and a
        sample of those where they agree) the exact list rebuilt from the path trace has length `hops`,
        ends at dst, starts with the stored next-hop, and the f64 product of the input edges along it
        equals the stored rate up to its own hops - 1 fp32 multiplications: <= (hops - 1) * 2^-24 * 1.001."""
    u = 2.0 ** -24
    same = src == dst
    assert bool((walk_len[same] == 0).all()) and bool((walk_len[~same] >= 1).all())
    solved = rate[src, dst].astype(np.float64)
    h = hops[src, dst].astype(np.int64)
    rel = np.abs(walk_prod - solved) / np.maximum(solved, 1e-300)
    bound = (walk_len.astype(np.int64) + h) * u
    worst = int(np.argmax((rel - bound)[~same]))
    assert bool((rel[~same] <= bound[~same]).all()), (float(rel[~same][worst]), float(bound[~same][worst]))
    differ = np.flatnonzero((walk_len != h) & ~same)
    agree = np.flatnonzero((walk_len == h) & ~same)[:2000]
    pick = np.concatenate([differ, agree])
    lists = exact_lists_of(src[pick], dst[pick])
    for q, i in enumerate(pick):
        s_, d_, path = int(src[i]), int(dst[i]), lists[q]
        assert len(path) == hops[s_, d_], (s_, d_, len(path), int(hops[s_, d_]), int(walk_len[i]))
        assert path[-1] == d_ and path[0] == nxt[s_, d_]
        p, cur = 1.0, s_
        for v in path:
            p *= float(rate0[cur, v])
            cur = v
        r = float(rate[s_, d_])
        assert abs(p - r) <= (len(path) - 1) * u * 1.001 * r, (s_, d_, p, r, len(path))
    return len(differ)
