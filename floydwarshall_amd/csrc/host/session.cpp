// session.cpp -- AppState (InSync/OutSync) and the request layer over the GPU engine.
//
//   AppState / blankState   /root/reference/src/lib/Types.hs:35-37, Utils.hs:16-17
//   updateRates             /root/reference/src/lib/ProcessRequests.hs:89-102
//   findBestRate/syncMatrix /root/reference/src/lib/ProcessRequests.hs:70-85
//   serveReq                /root/reference/src/lib/ProcessRequests.hs:31-63
//   Main.run                /root/reference/src/app/Main.hs:26-37
//
// floydWarshall (Algorithms.hs:19-20) = build_matrix on the host + fwx_matrix_solve on the GPU.
// The solved matrix stays resident in HBM (fwx_matrix) tagged with the rate-map version: queries
// read back one rate and one path, not the matrix, and a query that fails in optimum does not
// throw the solve away (the reference's lazy matrix costs nothing to rebuild; a GPU solve does).
#include <cstring>

#include "fwx.h"
#include "host_types.hpp"

namespace fwxh {

Session::Session(int device) : device_(device) {}

Session::~Session() { drop_device(); }

void Session::set_devices(const std::vector<int32_t> &devices, int32_t min_vertices)
{
    devices_ = devices;
    multi_from_ = min_vertices;
    drop_device();               // the next query re-creates the handle where it now belongs
    solved_version_ = ~0ull;
}

int Session::parts() const { return dev_ ? fwx_matrix_parts(dev_, nullptr) : 0; }

void Session::drop_device()
{
    if (dev_) {
        fwx_matrix_destroy(dev_);
        dev_ = nullptr;
    }
    rebuild_ = true;             // the kept input went with the handle
    patches_.clear();
}

bool Session::update_rates(int64_t time, const Vertex &src, const Vertex &dest, double fwd, double bkd)
{
    // :97-98  rateOutdated = lookup (src,dest) <&> ((< time) . snd); update if Nothing or True
    auto it = rates_.find(VertexPair(src, dest));
    const bool update_required = it == rates_.end() || it->second.second < time;
    if (!update_required) return false;
    // The dense matrix buildMatrix produces (Algorithms.hs:26-40) is a function of the key set and
    // of the rate VALUES only -- not of the timestamps.  An accepted update that re-quotes the same
    // two prices with a newer time (the common heartbeat of a feed) leaves it bit-identical, so the
    // solved matrix on the device is still floydWarshall of the new map: the visible state flips
    // to OutSync exactly as in the reference (:101-102), but the next query reuses the solve.
    auto same_bits = [](double a, double b) { return std::memcmp(&a, &b, sizeof(double)) == 0; };
    auto bk = rates_.find(VertexPair(dest, src));
    const bool matrix_unchanged = it != rates_.end() && bk != rates_.end() &&
                                  same_bits(it->second.first, fwd) && same_bits(bk->second.first, bkd);
    // :101-102 OutSync $ updateMap [((dest,src),(bkdR,time)), ((src,dest),(fwdR,time))] exRates
    rates_[VertexPair(dest, src)] = std::make_pair(bkd, time);
    rates_[VertexPair(src, dest)] = std::make_pair(fwd, time);
    in_sync_ = false;
    if (matrix_unchanged) return true;
    ++version_;
    if (!rebuild_) {
        // buildMatrix's output changes in exactly two entries if both vertices are known: (i,j) =
        // (fwd, [j]) and (j,i) = (bkd, [i]) (Algorithms.hs:36-37).  A new vertex renumbers the rows,
        // and two vertices of one currency take the 1.0 rule (:35, tested before the map): both go
        // through the full marshal.
        const auto si = vindex_.find(src), di = vindex_.find(dest);
        if (si == vindex_.end() || di == vindex_.end() || src.ccy == dest.ccy ||
            patches_.size() + 2 > kMaxPatches) {
            rebuild_ = true;
            patches_.clear();
        } else {
            const int64_t n = (int64_t)initial_.n(), i = si->second, j = di->second;
            const Patch two[2] = {{i * n + j, fwd, (int32_t)j, 1}, {j * n + i, bkd, (int32_t)i, 1}};
            for (const Patch &p : two) {
                initial_.rate[(size_t)p.idx] = p.rate;
                initial_.next[(size_t)p.idx] = p.next;
                initial_.hops[(size_t)p.idx] = p.hops;
                patches_.push_back(p);
            }
        }
    }
    return true;
}

int Session::ensure_solved()
{
    if (solved_version_ == version_) return FWX_OK;
    DenseMatrix &m = initial_;
    if (!rebuild_ && dev_ && !patches_.empty()) {
        // same vertices, a few changed entries: patch the input kept on the device and solve it
        std::vector<int64_t> idx;
        std::vector<double> rate;
        std::vector<int32_t> next, hops;
        for (const Patch &p : patches_) {
            idx.push_back(p.idx);
            rate.push_back(p.rate); next.push_back(p.next); hops.push_back(p.hops);
        }
        // runAlgo on the patched input: from the last stored state the changed entries cannot have
        // influenced (fwx_matrix_resolve; bit-identical to runAlgo 0), else from pivot 0
        int32_t started = 0;
        const int rc = fwx_matrix_resolve(dev_, (int32_t)idx.size(), idx.data(), rate.data(), next.data(),
                                          dev_hops_ ? hops.data() : nullptr, nullptr, &started);
        if (!rc) {
            patches_.clear();
            solved_version_ = version_;
            ++solves_;
            ++patched_solves_;
            if (started > 0) {
                ++resumed_solves_;
                resumed_pivots_ += started;
            }
            return FWX_OK;
        }
        drop_device();           // whatever went wrong: marshal from scratch below
    }
    build_matrix_into(rates_, m);
    // The device handle (matrix, pristine copies, log arrays) is kept while the vertex count stays
    // the same -- a rate update between known vertices, the common case -- and only re-uploaded.
    // (any vertex count: the handles pad odd orders on the device themselves, fwx.h fwx_engine)
    const bool want_multi = !devices_.empty() && m.n() >= multi_from_;
    if (dev_ && (dev_n_ != m.n() || dev_multi_ != want_multi)) drop_device();
    vertices_ = m.vertices;
    if (m.n() > 0) {
        // The solve keeps the path trace (fwx_matrix_enable_path_log), from which
        // fwx_matrix_query_exact rebuilds the reference's `_path` lists -- under exact ties (the
        // 1.0 edges of Algorithms.hs:35 make them common) the list the reference stored can be a
        // longer route than the one the next-hops describe.  Requests need rates and paths, not
        // `hops`: from kFusedFrom vertices on the resident matrix carries none (one n x n array, its
        // upload and its panel traffic less per re-solve); solved_matrix() computes them on demand.
        int rc = FWX_OK;
        dev_hops_ = m.n() < kFusedFrom;
        if (!dev_) {
            // one floydWarshall call, the whole node behind it: a partitioned handle is an
            // ordinary fwx_matrix (same upload / solve / query_exact below)
            rc = want_multi ? fwx_matrix_create_multi(&dev_, m.n(), FWX_F64, 1, dev_hops_ ? 1 : 0,
                                                      (int32_t)devices_.size(), devices_.data(),
                                                      FWX_XCHG_AUTO)
                            : fwx_matrix_create(&dev_, m.n(), FWX_F64, 1, dev_hops_ ? 1 : 0,
                                                devices_.empty() ? device_ : devices_[0]);
            if (rc) return rc;
            dev_multi_ = want_multi;
            dev_n_ = m.n();
            if ((rc = fwx_matrix_enable_path_log(dev_)) || (rc = fwx_matrix_keep_input(dev_))) {
                drop_device();
                return rc;
            }
            // State checkpoints + panels, so that a re-solve after a price change can resume (f3).
            // Resuming is an optimisation, never a requirement: the count is cut to what fits into half
            // of the device's free memory (a checkpoint is a copy of every array -- 24 B per entry for
            // this traced f64 matrix -- and the panels another ~2.5 arrays: 7 checkpoints of a 16384-vertex
            // market would be 45 GB), and where nothing fits, the allocation fails anyway, or the handle
            // cannot resume at all (small, partitions: FWX_ERR_UNSUPPORTED) the session carries on with
            // full solves from the patched input.
            dev_checkpoints_ = 0;
            int32_t want = checkpoints_;
            if (want > 0) {
                uint64_t free_b = 0, total_b = 0, need = 0;
                if (fwx_device_memory(devices_.empty() ? device_ : devices_[0], &free_b, &total_b) == FWX_OK)
                    while (want > 0 && fwx_matrix_resume_bytes(dev_, want, &need) == FWX_OK && need > free_b / 2)
                        --want;
            }
            if (want > 0) {
                const int placed = fwx_matrix_enable_resume(dev_, want);
                if (placed > 0) dev_checkpoints_ = placed;        // any failure: no resume, the solve goes on
            }
        }
        rc = fwx_matrix_upload(dev_, m.rate.data(), m.next.data(), dev_hops_ ? m.hops.data() : nullptr);
        if (rc || (rc = fwx_matrix_solve(dev_, nullptr))) {         // runAlgo 0, on the GPU
            drop_device();
            return rc;
        }
    } else {
        drop_device();
    }
    vindex_.clear();
    for (int32_t i = 0; i < m.n(); ++i) vindex_[m.vertices[(size_t)i]] = i;
    rebuild_ = m.n() == 0;       // from here on updates between known vertices are patched in
    patches_.clear();
    solved_version_ = version_;
    ++solves_;
    return FWX_OK;
}

OptimumResult Session::find_best_rate(const Vertex &src, const Vertex &dest)
{
    OptimumResult res;
    // :77-79  syncMatrix: OutSync -> floydWarshall, put InSync
    const bool was_in_sync = in_sync_;
    const int rc = ensure_solved();
    if (rc) {
        res.status = rc;
        res.error = std::string("engine: ") + fwx_strerror(rc);
        return res;
    }
    // :80  optimum src dest matrix -- index lookups on the host, entry + path from the device
    const int32_t n = (int32_t)vertices_.size();
    auto idx = [&](const Vertex &v) {
        for (int32_t i = 0; i < n; ++i)
            if (vertices_[i] == v) return i;
        return (int32_t)-1;
    };
    const int32_t s = idx(src);
    if (s < 0) { res.error = src.show() + " is not entered before"; }
    const int32_t d = s < 0 ? -1 : idx(dest);
    if (s >= 0 && d < 0) { res.error = dest.show() + " is not entered before"; }
    if (s >= 0 && d >= 0) {
        // the reference's `_path`, exactly (path trace); the buffer grows for arbitrage blow-ups
        std::vector<int32_t> path((size_t)(4 * n > 64 ? 4 * n : 64));
        double rate = 0.0;
        int len = fwx_matrix_query_exact(dev_, s, d, &rate, path.data(), (int32_t)path.size());
        while (len == FWX_ERR_CAPACITY && path.size() < ((size_t)1 << 24)) {
            path.resize(path.size() * 8);
            len = fwx_matrix_query_exact(dev_, s, d, &rate, path.data(), (int32_t)path.size());
        }
        if (len < 0) {
            res.status = len;
            res.error = std::string("engine: ") + fwx_strerror(len);
            return res;
        }
        if (len == 0) {
            res.error = "There is no exchange between " + src.show() + " and " + dest.show();
        } else {
            res.ok = true;
            res.rate = rate;
            res.start = src;
            for (int i = 0; i < len; ++i) res.path.push_back(vertices_[path[i]]);
        }
    }
    // The reference runs in RWST .. (Either e): when optimum fails the `put (InSync ..)` of :79
    // is rolled back with everything else, so the visible state only advances on success.
    in_sync_ = res.ok ? true : was_in_sync;
    return res;
}

int Session::solved_matrix(DenseMatrix &out)
{
    const int rc = ensure_solved();
    if (rc) return rc;
    in_sync_ = true;
    out.vertices = vertices_;
    const size_t n = vertices_.size();
    out.rate.assign(n * n, 0.0);
    out.next.assign(n * n, -1);
    out.hops.assign(n * n, 0);
    if (n == 0) return FWX_OK;
    if (dev_hops_)
        return fwx_matrix_download(dev_, out.rate.data(), out.next.data(), out.hops.data());
    // large matrix: the resident solve carries no `hops` (length _path); this rarely used call
    // solves the initial matrix once more with them, on the engine that carries hops
    out.rate = initial_.rate;
    out.next = initial_.next;
    out.hops = initial_.hops;
    fwx_opts o;
    memset(&o, 0, sizeof(o));
    o.struct_size = sizeof(o);
    o.device = device_;
    return fwx_solve_f64((int32_t)n, out.rate.data(), out.next.data(), out.hops.data(), &o);
}

std::vector<std::string> Session::serve_line(const std::string &line)
{
    std::vector<std::string> err, res;
    // serveReq: catchError updateRatesM (\err1 -> ...)            ProcessRequests.hs:34-35
    ParsedRates pr;
    std::string e1;
    if (parse_rates(line, pr, e1)) {
        update_rates(pr.time, pr.src, pr.dest, pr.fwd, pr.bkd);
        // :46-50 one line per stored rate, ascending key order
        for (const auto &kv : rates_)
            res.push_back(kv.first.first.show() + " -- " + show_double(kv.second.first) + " " +
                          show_utctime(kv.second.second) + " --> " + kv.first.second.show());
    } else {
        // :36-38
        err.push_back(e1);
        err.push_back("Invalid request to update rates, probably a request for best rate");
        // :39 catchError findBestRateM (\err2 -> tell err2)
        Vertex src, dest;
        std::string e2;
        if (!parse_exch_pair(line, src, dest, e2)) {
            err.push_back(e2);
        } else {
            OptimumResult r = find_best_rate(src, dest);
            if (!r.ok) {
                err.push_back(r.error);
            } else {
                // presentRateEntry :53-63
                const Vertex &last = r.path.back();
                res.push_back("BEST_RATES_BEGIN " + r.start.exch + " " + r.start.ccy + " " +
                              last.exch + " " + last.ccy + " " + show_double(r.rate));
                res.push_back(r.start.show());
                for (const auto &v : r.path) res.push_back(v.show());
                res.push_back("BEST_RATES_END");
            }
        }
    }
    // Main.run: Main.hs:34-37
    std::vector<std::string> out;
    if (res.empty()) {
        out = err;
        out.push_back("You neither enter exchange rates or request best rate, please enter a valid input\n");
    } else {
        out = res;
        out.push_back("");
    }
    return out;
}

}  // namespace fwxh
