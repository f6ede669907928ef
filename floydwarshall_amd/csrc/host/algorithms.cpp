// algorithms.cpp -- host mirror of the reference's Algorithms module on the dense form.
//
//   buildMatrix  /root/reference/src/lib/Algorithms.hs:26-40
//   optimum      /root/reference/src/lib/Algorithms.hs:65-78
// (floydWarshall = runAlgo 0 . buildMatrix, :19-20, is Session::ensure_solved in session.cpp:
//  buildMatrix here, runAlgo on the GPU through fwx_matrix_solve.)
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "host_types.hpp"

namespace fwxh {

DenseMatrix build_matrix(const ExchRateTimes &rates)
{
    DenseMatrix m;
    build_matrix_into(rates, m);
    return m;
}

// Same, into a caller-kept DenseMatrix: a host that rebuilds after every rate update reuses the
// three n x n host arrays instead of faulting in fresh pages each time.
void build_matrix_into(const ExchRateTimes &rates, DenseMatrix &m)
{
    // :29  vertices = sort . nub $ keys >>= \(k1,k2) -> [k1,k2]
    std::set<Vertex> vs;
    for (const auto &kv : rates) {
        vs.insert(kv.first.first);
        vs.insert(kv.first.second);
    }
    m.vertices.assign(vs.begin(), vs.end());
    const size_t n = m.vertices.size();
    m.rate.assign(n * n, 0.0);       // isolatedEntry: rate 0.0, path [] (Utils.hs:13-14)
    m.next.assign(n * n, -1);
    m.hops.assign(n * n, 0);
    // The reference evaluates, per entry, :34 (i == j) then :35 (same currency) then :36-38 (map
    // lookup): N^2 lookups.  Same result in O(E log V + sum of group^2): scatter the map entries
    // first, then let the same-currency rule overwrite (it is tested BEFORE the lookup, so it wins).
    std::map<Vertex, size_t> index;
    for (size_t i = 0; i < n; ++i) index[m.vertices[i]] = i;
    for (const auto &kv : rates) {                                        // :36-37
        const size_t i = index[kv.first.first], j = index[kv.first.second];
        if (i == j) continue;                                             // :34
        m.rate[i * n + j] = kv.second.first;
        m.next[i * n + j] = (int32_t)j;                                   // path = [vtxJ]
        m.hops[i * n + j] = 1;
    }
    std::map<std::string, std::vector<size_t>> by_ccy;
    for (size_t i = 0; i < n; ++i) by_ccy[m.vertices[i].ccy].push_back(i);
    for (const auto &g : by_ccy)                                          // :35
        for (size_t i : g.second)
            for (size_t j : g.second) {
                if (i == j) continue;
                m.rate[i * n + j] = 1.0;
                m.next[i * n + j] = (int32_t)j;
                m.hops[i * n + j] = 1;
            }
}

OptimumResult optimum_dense(const std::vector<Vertex> &vertices, int32_t n_cols, const double *rate,
                            const int32_t *next, const Vertex &src, const Vertex &dest)
{
    OptimumResult res;
    const int32_t n_rows = (int32_t)vertices.size();
    // :70-71  traverse ((fmap _start) . (!? 0)) matrix -- Nothing as soon as one row is empty.
    // A matrix with zero rows traverses to Just [] and falls through to the index lookups.
    if (n_rows > 0 && n_cols == 0) {
        res.error = "The matrix is empty";
        return res;
    }
    auto vertice_idx = [&](const Vertex &v) -> int32_t {                  // :77 elemIndex
        for (int32_t i = 0; i < n_rows; ++i)
            if (vertices[i] == v) return i;
        return -1;
    };
    const int32_t s = vertice_idx(src);
    if (s < 0) {                                                           // :72
        res.error = src.show() + " is not entered before";
        return res;
    }
    const int32_t d = vertice_idx(dest);
    if (d < 0) {                                                           // :73
        res.error = dest.show() + " is not entered before";
        return res;
    }
    const std::string not_reachable =
        "There is no exchange between " + src.show() + " and " + dest.show();   // :78
    if (d >= n_cols || next[(size_t)s * n_cols + d] < 0) {                 // :74-75 null _path
        res.error = not_reachable;
        return res;
    }
    // `_path` = follow head-of-path from src until dest (index form of the reference's list)
    int32_t cur = s, len = 0;
    while (cur != d || len == 0) {
        const int32_t nx = next[(size_t)cur * n_cols + d];
        if (nx < 0 || nx >= n_rows || len >= n_rows) {
            res.error = "next-hop walk does not reach the destination (arbitrage cycle)";
            res.status = -5;  // FWX_ERR_CYCLE
            return res;
        }
        res.path.push_back(vertices[nx]);
        cur = nx;
        ++len;
    }
    res.ok = true;
    res.rate = rate[(size_t)s * n_cols + d];
    res.start = src;
    return res;
}

}  // namespace fwxh
