// host_api.cpp -- extern "C" surface of the host mirror (include/fwx_host.h).
#include <cstring>
#include <new>

#include "fwx.h"
#include "fwx_guard.h"
#include "fwx_host.h"
#include "host_types.hpp"

using namespace fwxh;

struct fwxh_session {
    Session impl;
    explicit fwxh_session(int device) : impl(device) {}
};

namespace {

int put(const std::string &s, char *buf, size_t cap)
{
    if (!buf) return cap == 0 ? 0 : FWX_ERR_INVALID;
    if (s.size() + 1 > cap) {
        if (cap) buf[0] = 0;
        return FWX_ERR_CAPACITY;
    }
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

int emit_result(const OptimumResult &r, double *rate_out, char *path_buf, size_t path_cap,
                char *err_buf, size_t err_cap)
{
    if (!r.ok) {
        if (err_buf && err_cap) put(r.error, err_buf, err_cap);
        return r.status ? r.status : FWXH_ERR_ALGO;
    }
    if (rate_out) *rate_out = r.rate;
    std::string p = r.start.show() + "\n";
    for (const auto &v : r.path) p += v.show() + "\n";
    if (path_buf) {
        const int rc = put(p, path_buf, path_cap);
        if (rc) return rc;
    }
    return (int)r.path.size();
}

}  // namespace

extern "C" {

int fwxh_session_create(fwxh_session **out, int32_t device)
{
    if (!out) return FWX_ERR_INVALID;
    *out = nullptr;
    return fwxi::guarded([&]() -> int {
        *out = new (std::nothrow) fwxh_session(device);
        return *out ? FWX_OK : FWX_ERR_OOM;
    });
}

int fwxh_session_destroy(fwxh_session *s)
{
    return fwxi::guarded([&]() -> int {
        delete s;
        return FWX_OK;
    });
}

int fwxh_session_set_devices(fwxh_session *s, int32_t n_parts, const int32_t *devices, int32_t min_vertices)
{
    if (!s || n_parts < 0 || n_parts > FWX_MAX_PARTS || (n_parts > 0 && !devices) || min_vertices < 0)
        return FWX_ERR_INVALID;
    return fwxi::guarded([&]() -> int {
        fwxi::fail_point();
        s->impl.set_devices(std::vector<int32_t>(devices, devices + n_parts), min_vertices);
        return FWX_OK;
    });
}

int32_t fwxh_session_parts(const fwxh_session *s) { return s ? s->impl.parts() : -1; }

int fwxh_session_state(const fwxh_session *s) { return s ? s->impl.state() : FWX_ERR_INVALID; }

int64_t fwxh_session_solves(const fwxh_session *s) { return s ? s->impl.solves() : -1; }

int64_t fwxh_session_patched_solves(const fwxh_session *s) { return s ? s->impl.patched_solves() : -1; }

int64_t fwxh_session_resumed_solves(const fwxh_session *s) { return s ? s->impl.resumed_solves() : -1; }

int64_t fwxh_session_resumed_pivots(const fwxh_session *s) { return s ? s->impl.resumed_pivots() : -1; }

int32_t fwxh_session_checkpoints_kept(const fwxh_session *s) { return s ? s->impl.checkpoints_kept() : -1; }

int fwxh_session_set_checkpoints(fwxh_session *s, int32_t checkpoints)
{
    if (!s || checkpoints < 0 || checkpoints > FWX_MAX_CHECKPOINTS) return FWX_ERR_INVALID;
    return fwxi::guarded([&]() -> int {
        s->impl.set_checkpoints(checkpoints);
        return FWX_OK;
    });
}

int32_t fwxh_session_rate_count(const fwxh_session *s)
{
    return s ? (int32_t)s->impl.rates().size() : -1;
}

int fwxh_update_rates(fwxh_session *s, int64_t posix_seconds, const char *exch, const char *src_ccy,
                      const char *dst_ccy, double fwd_rate, double bkd_rate)
{
    if (!s || !exch || !src_ccy || !dst_ccy) return FWX_ERR_INVALID;
    try {
        return s->impl.update_rates(posix_seconds, Vertex{exch, src_ccy}, Vertex{exch, dst_ccy},
                                    fwd_rate, bkd_rate) ? 1 : 0;
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_build_matrix(const fwxh_session *s, int32_t *n_out, double *rate, int32_t *next,
                      char *vertex_buf, size_t vertex_cap)
{
    if (!s || !n_out) return FWX_ERR_INVALID;
    try {
        DenseMatrix m = build_matrix(s->impl.rates());
        *n_out = m.n();
        const size_t nn = (size_t)m.n() * m.n();
        if (rate && nn) memcpy(rate, m.rate.data(), nn * sizeof(double));
        if (next && nn) memcpy(next, m.next.data(), nn * sizeof(int32_t));
        if (vertex_buf) {
            std::string v;
            for (const auto &x : m.vertices) v += x.exch + " " + x.ccy + "\n";
            return put(v, vertex_buf, vertex_cap);
        }
        return FWX_OK;
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_find_best_rate(fwxh_session *s, const char *src_exch, const char *src_ccy,
                        const char *dst_exch, const char *dst_ccy, double *rate_out,
                        char *path_buf, size_t path_cap, char *err_buf, size_t err_cap)
{
    if (!s || !src_exch || !src_ccy || !dst_exch || !dst_ccy) return FWX_ERR_INVALID;
    try {
        OptimumResult r = s->impl.find_best_rate(Vertex{src_exch, src_ccy}, Vertex{dst_exch, dst_ccy});
        return emit_result(r, rate_out, path_buf, path_cap, err_buf, err_cap);
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_solved_matrix(fwxh_session *s, int32_t *n_out, double *rate, int32_t *next, int32_t *hops)
{
    if (!s || !n_out) return FWX_ERR_INVALID;
    try {
        DenseMatrix m;
        const int rc = s->impl.solved_matrix(m);
        if (rc) return rc;
        *n_out = m.n();
        const size_t nn = (size_t)m.n() * m.n();
        if (rate && nn) memcpy(rate, m.rate.data(), nn * sizeof(double));
        if (next && nn) memcpy(next, m.next.data(), nn * sizeof(int32_t));
        if (hops && nn) memcpy(hops, m.hops.data(), nn * sizeof(int32_t));
        return FWX_OK;
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_optimum_dense(int32_t n_rows, int32_t n_cols, const char *const *exch,
                       const char *const *ccy, const double *rate, const int32_t *next,
                       const char *src_exch, const char *src_ccy, const char *dst_exch,
                       const char *dst_ccy, double *rate_out, char *path_buf, size_t path_cap,
                       char *err_buf, size_t err_cap)
{
    if (n_rows < 0 || n_cols < 0 || !src_exch || !src_ccy || !dst_exch || !dst_ccy)
        return FWX_ERR_INVALID;
    if (n_rows > 0 && (!exch || !ccy)) return FWX_ERR_INVALID;
    if (n_rows > 0 && n_cols > 0 && (!rate || !next)) return FWX_ERR_INVALID;
    try {
        std::vector<Vertex> vs;
        for (int32_t i = 0; i < n_rows; ++i) vs.push_back(Vertex{exch[i], ccy[i]});
        OptimumResult r = optimum_dense(vs, n_cols, rate, next, Vertex{src_exch, src_ccy},
                                        Vertex{dst_exch, dst_ccy});
        return emit_result(r, rate_out, path_buf, path_cap, err_buf, err_cap);
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_parse_rates(const char *line, int64_t *posix_seconds, char *exch, char *src_ccy,
                     char *dst_ccy, size_t cap, double *fwd, double *bkd, char *err_buf,
                     size_t err_cap)
{
    if (!line) return FWX_ERR_INVALID;
    try {
        ParsedRates pr;
        std::string err;
        if (!parse_rates(line, pr, err)) {
            if (err_buf && err_cap) put(err, err_buf, err_cap);
            return FWXH_ERR_PARSE;
        }
        if (posix_seconds) *posix_seconds = pr.time;
        if (fwd) *fwd = pr.fwd;
        if (bkd) *bkd = pr.bkd;
        int rc = 0;
        if (exch && (rc = put(pr.src.exch, exch, cap))) return rc;
        if (src_ccy && (rc = put(pr.src.ccy, src_ccy, cap))) return rc;
        if (dst_ccy && (rc = put(pr.dest.ccy, dst_ccy, cap))) return rc;
        return FWX_OK;
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_parse_exch_pair(const char *line, char *src_exch, char *src_ccy, char *dst_exch,
                         char *dst_ccy, size_t cap, char *err_buf, size_t err_cap)
{
    if (!line) return FWX_ERR_INVALID;
    try {
        Vertex s, d;
        std::string err;
        if (!parse_exch_pair(line, s, d, err)) {
            if (err_buf && err_cap) put(err, err_buf, err_cap);
            return FWXH_ERR_PARSE;
        }
        int rc = 0;
        if (src_exch && (rc = put(s.exch, src_exch, cap))) return rc;
        if (src_ccy && (rc = put(s.ccy, src_ccy, cap))) return rc;
        if (dst_exch && (rc = put(d.exch, dst_exch, cap))) return rc;
        if (dst_ccy && (rc = put(d.ccy, dst_ccy, cap))) return rc;
        return FWX_OK;
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_serve_line(fwxh_session *s, const char *line, char *out, size_t out_cap)
{
    if (!s || !line) return FWX_ERR_INVALID;
    try {
        std::string all;
        for (const auto &l : s->impl.serve_line(line)) all += l + "\n";   // putStrLn each
        return put(all, out, out_cap);
    } catch (...) { return FWX_ERR_OOM; }
}

int fwxh_show_double(double x, char *out, size_t cap)
{
    try { return put(show_double(x), out, cap); } catch (...) { return FWX_ERR_OOM; }
}

}  // extern "C"
