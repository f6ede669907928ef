// algorithms.cpp -- host mirror of the reference's Algorithms module on the dense form.
//
//   buildMatrix  /root/reference/src/lib/Algorithms.hs:26-40
//   optimum      /root/reference/src/lib/Algorithms.hs:65-78
// (floydWarshall = runAlgo 0 . buildMatrix, :19-20, is Session::ensure_solved in session.cpp:
//  buildMatrix here, runAlgo on the GPU through fwx_matrix_solve.)
#include <algorithm>
#include <set>

#include "host_types.hpp"

namespace fwxh {

DenseMatrix build_matrix(const ExchRateTimes &rates)
{
    DenseMatrix m;
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
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;                                         // :34
            const Vertex &vi = m.vertices[i], &vj = m.vertices[j];
            double r;
            if (vi.ccy == vj.ccy) {                                       // :35 before the lookup
                r = 1.0;
            } else {
                auto it = rates.find(VertexPair(vi, vj));                 // :36
                if (it == rates.end()) continue;                          // :38
                r = it->second.first;                                     // :37
            }
            m.rate[i * n + j] = r;
            m.next[i * n + j] = (int32_t)j;                               // path = [vtxJ]
            m.hops[i * n + j] = 1;
        }
    }
    return m;
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
