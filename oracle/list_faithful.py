"""List-faithful CPU restatement of the reference's Algorithms module (ORACLE -- test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (floydwarshall_amd/) never does.

This file restates, entry for entry and with whole `_path` lists, what the Haskell reference
computes, so that the dense next-hop form used by the engine (and by oracle/fw_oracle.c) can be
shown equal to it:

    buildMatrix    /root/reference/src/lib/Algorithms.hs:26-40
    runAlgo        /root/reference/src/lib/Algorithms.hs:42-61
    floydWarshall  /root/reference/src/lib/Algorithms.hs:19-20
    optimum        /root/reference/src/lib/Algorithms.hs:65-78
    Vertex Ord/Show  /root/reference/src/lib/Types.hs:13-20
    isolatedEntry  /root/reference/src/lib/Utils.hs:13-14

Pure-Python loops: small N only (the reference's own tests stop at 4x4).

A Vertex is a tuple (exch, ccy) -- tuple ordering is the derived `Ord` (exch first, then ccy).
A RateEntry is a tuple (best_rate, start_vertex, path) with path a tuple of vertices.
A Matrix is a list of rows, each a list of RateEntry.
"""
import numpy as np


def show_vertex(v):
    """`Show Vertex` -- Types.hs:19-20: "(EXCH, CCY)"."""
    return "(" + v[0] + ", " + v[1] + ")"


def _mul(a, b, dtype):
    """One IEEE multiply in the working precision (binary64 in the reference, Types.hs:26)."""
    if dtype == np.float64:
        return float(a) * float(b)
    with np.errstate(all="ignore"):
        return dtype(dtype(a) * dtype(b))


def build_matrix(ex_rates, dtype=np.float64):
    """Algorithms.hs:26-40.  ex_rates: dict {(vertex_src, vertex_dst): rate}."""
    # :29  vertices = sort . nub $ keys >>= \(k1,k2) -> [k1,k2]
    vertices = sorted({v for key in ex_rates for v in key})
    n = len(vertices)
    matrix = []
    for i in range(n):
        vtx_i = vertices[i]
        isolated = (dtype(0.0), vtx_i, ())                       # Utils.hs:13-14
        row = []
        for j in range(n):
            vtx_j = vertices[j]
            if i == j:                                           # :34
                row.append(isolated)
            elif vtx_i[1] == vtx_j[1]:                           # :35 same currency, checked first
                row.append((dtype(1.0), vtx_i, (vtx_j,)))
            elif (vtx_i, vtx_j) in ex_rates:                     # :36-37
                row.append((dtype(ex_rates[(vtx_i, vtx_j)]), vtx_i, (vtx_j,)))
            else:                                                # :38
                row.append(isolated)
        matrix.append(row)
    return matrix


def run_algo(matrix, dtype=np.float64, k_begin=0, k_end=None):
    """Algorithms.hs:42-61: a NEW matrix per k, every operand read from the previous one."""
    n = len(matrix)
    k_end = n if k_end is None else k_end
    for k in range(k_begin, k_end):                              # :44
        new_matrix = []
        for i in range(n):
            if i == k:                                           # :50
                new_matrix.append(matrix[k])
                continue
            new_row = []
            for j in range(n):
                orig = matrix[i][j]
                if j == i or j == k:                             # :54
                    new_row.append(orig)
                    continue
                ik_rate, _, ik_path = matrix[i][k]               # :59
                kj_rate, _, kj_path = matrix[k][j]               # :60
                new_rate = _mul(ik_rate, kj_rate, dtype)         # :61
                if orig[0] < new_rate:                           # :55 strict
                    new_row.append((new_rate, orig[1], ik_path + kj_path))
                else:
                    new_row.append(orig)
            new_matrix.append(new_row)
        matrix = new_matrix
    return matrix


def floyd_warshall(ex_rates, dtype=np.float64):
    """Algorithms.hs:19-20: runAlgo 0 . buildMatrix."""
    return run_algo(build_matrix(ex_rates, dtype), dtype)


def optimum(src, dest, matrix):
    """Algorithms.hs:65-78.  Returns ("ok", entry) or ("err", message)."""
    # :70  traverse ((fmap _start) . (!? 0)) matrix  -- Nothing if ANY row is empty
    starts = []
    for row in matrix:
        if len(row) == 0:
            return ("err", "The matrix is empty")                # :71
        starts.append(row[0][1])

    def vertice_idx(v):                                          # :77
        return starts.index(v) if v in starts else None

    src_idx = vertice_idx(src)
    if src_idx is None:                                          # :72
        return ("err", show_vertex(src) + " is not entered before")
    dest_idx = vertice_idx(dest)
    if dest_idx is None:                                         # :73
        return ("err", show_vertex(dest) + " is not entered before")
    not_reachable = ("There is no exchange between " + show_vertex(src) + " and "
                     + show_vertex(dest))                        # :78
    if dest_idx >= len(matrix[src_idx]):                         # :74 (!?)
        return ("err", not_reachable)
    entry = matrix[src_idx][dest_idx]
    if len(entry[2]) == 0:                                       # :75
        return ("err", not_reachable)
    return ("ok", entry)


# ------------------------------------------------------------------------------------------------
# Conversions between the list form and the dense SoA form
# ------------------------------------------------------------------------------------------------

def to_dense(matrix, dtype=np.float64):
    """(vertices, rate[n,n], next[n,n] = index of head _path or -1, hops[n,n] = length _path)."""
    n = len(matrix)
    vertices = [row[0][1] for row in matrix]
    index = {v: i for i, v in enumerate(vertices)}
    rate = np.zeros((n, n), dtype=dtype)
    nxt = np.full((n, n), -1, dtype=np.int32)
    hops = np.zeros((n, n), dtype=np.int32)
    for i in range(n):
        for j in range(n):
            r, _, path = matrix[i][j]
            rate[i, j] = r
            hops[i, j] = len(path)
            if path:
                nxt[i, j] = index[path[0]]
    return vertices, rate, nxt, hops


def from_dense(vertices, rate, nxt=None):
    """Initial dense arrays -> list form with single-hop paths (inverse of to_dense at k=0)."""
    n = len(vertices)
    matrix = []
    for i in range(n):
        row = []
        for j in range(n):
            if nxt is None:
                path = (vertices[j],) if i != j else ()
            else:
                path = (vertices[int(nxt[i, j])],) if nxt[i, j] >= 0 else ()
            row.append((rate[i, j], vertices[i], path))
        matrix.append(row)
    return matrix


def path_indices(matrix):
    """Per entry, the `_path` as a tuple of vertex indices (the form AlgorithmsTest.hs:55-58 uses)."""
    vertices = [row[0][1] for row in matrix]
    index = {v: i for i, v in enumerate(vertices)}
    return [[tuple(index[v] for v in e[2]) for e in row] for row in matrix]


__all__ = ["show_vertex", "build_matrix", "run_algo", "floyd_warshall", "optimum", "to_dense",
           "from_dense", "path_indices"]
