// fwx_kernels.h -- internal launch interface between the C ABI (fwx_api.hip) and the kernels.
#ifndef FWX_KERNELS_H
#define FWX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// Must equal FWX_UPDATE_SHARDS of include/fwx.h (power of two).
#define FWX_UPDATE_SHARDS_K 256

namespace fwx {

// Path trace for exact `_path` reconstruction (Algorithms.hs:55 concatenates the time-k paths of
// (i,k) and (k,j) whenever (i,j) improves).
struct PathLog {
    // What the reference's `_path` lists (Algorithms.hs:55) need, without storing a single list or
    // update record: three n x n int32 matrices.
    //   last[i][j]    pivot of the newest successful relaxation of (i,j) in the whole solve, -1 = never
    //   at_col[i][j]  the same, but as it stood at the START of step j  (entry (i,j) as pivot-COLUMN operand)
    //   at_row[i][j]  ...                 at the START of step i        (entry (i,j) as pivot-ROW operand)
    // path(i,j) = path_q(i,q) ++ path_q(q,j) with q = last[i][j]; path_q(i,q) needs the newest update of
    // (i,q) before step q = at_col[i][q], path_q(q,j) needs at_row[q][j]; the recursion only ever asks
    // for an entry "as of the step named by one of its own indices", so these three suffice.
    // Column k and row k of `last` are not modified during step k: the snapshots are plain copies.
    int32_t *last = nullptr;       // (null by default: a PathLog that was never filled in means "no trace")
    int32_t *at_col = nullptr;
    int32_t *at_row = nullptr;
};

template <typename T> struct RelaxArgs {
    T *rate;                       // slab: rows x n
    int32_t *next;                 // or nullptr
    int32_t *hops;                 // or nullptr
    const T *prow;                 // pivot row k at the start of step k (n elements)
    const int32_t *phops;          // its hops row (iff hops)
    const int32_t *pnext = nullptr;   // its next-hop row, or nullptr.  Algorithms.hs:55 concatenates
                                   // ikPath ++ kjPath: the head is next[i][k] unless ikPath is EMPTY
                                   // (next[i][k] < 0), then it is next[k][j].  nullptr: the caller
                                   // vouches that a winning product never has an empty ikPath (true
                                   // on the reference's domain, see fwx.h "Domain")
    int rows, n, row0, k, flip;
    int skip_lo = 0, skip_hi = 0;  // slab rows [skip_lo, skip_hi) are left alone (multiples of 4):
                                   //   a look-ahead launch has already relaxed them
    unsigned long long *updates;   // FWX_UPDATE_SHARDS_K counters or nullptr
    PathLog plog;                  // plog.last == nullptr: no tracing (per-k engine: the slab must be the
                                   // whole matrix when tracing, the three matrices are indexed by
                                   // global row)
};

template <typename T> hipError_t launch_relax(const RelaxArgs<T> &a, hipStream_t s);

// Whole solve (pivots [k_begin,k_end)) of an n <= FWX_SMALL_N matrix in one single-workgroup launch.
#define FWX_SMALL_N 128
template <typename T>
hipError_t launch_small_solve(T *rate, int32_t *next, int32_t *hops, int n, int k_begin, int k_end,
                              unsigned long long *updates, PathLog plog, hipStream_t s);

template <typename T>
hipError_t launch_snapshot_row(T *dst, const T *src, int32_t *hdst, const int32_t *hsrc, int n,
                               hipStream_t s);

// ---- fused engine: FWX_FUSED_B pivots per launch from time-k snapshots ------------------------
#define FWX_FUSED_B 64

template <typename T> struct FusedArgs {
    T *rate;               // slab: rows x n (all rows are relaxed, pivot rows included)
    int32_t *next;         // or nullptr
    int rows, n, row0;
    int k0, bt;            // pivots [k0, k0+bt), bt <= FWX_FUSED_B
    const T *w;            // snapshot panel: w[t*n + j] = row k0+t at time k0+t
    T *ct;                 // scratch bt x rows: ct[t*rows + i] = column k0+t at time k0+t (NaN if i==k)
    int32_t *cnt;          // scratch bt x rows (iff next)
    int32_t *hops = nullptr;      // slab rows x n, or nullptr (needs next): hops' = hops[i][k] + hops[k][j]
    const int32_t *wh = nullptr;  // hops panel: wh[t*n + j] = hops of row k0+t at time k0+t (iff hops)
    int32_t *cht = nullptr;       // scratch bt x rows (iff hops): hops of column k0+t at time k0+t
    int ct_ld;             // leading dimension of ct / cnt (>= rows; a multiple of 4 keeps the
                           // staging loads 16-byte aligned)
    unsigned long long *updates;
    bool nonneg;           // caller verified the domain (fwx.h "Domain"): every matrix entry is >= +0
                           // and not NaN and, if next is carried, no positive rate lacks a path:
                           // the max-form kernels may run (f32: rates only, or rates + next + trace)
    PathLog plog = PathLog();   // path trace (needs next): rows x n like rate / next, LOCAL rows
    bool side = false;     // launch_fused_panels: launched beside a main launch (may pick a form that fits its holes);
                           // launch_fused_main: a launch of the look-ahead chain beside a main launch (the next
                           // blocks' rows and columns) -- its waves run at raised priority (s_setprio 2; the
                           // panel kernels always run at 3), because the chain, not the sweep, bounds mid sizes
};

// Domain check (fwx.h "Domain").  *flag is a device int preset to 3; bit 0 is cleared if any of the
// `count` rates is negative, -0 or NaN; bit 1 is cleared if `next` is given and some entry has a
// non-zero rate with next < 0 (a positive rate without a path).
hipError_t launch_nonneg_check(const float *rate, const int32_t *next, size_t count, int *flag,
                               hipStream_t s);
hipError_t launch_nonneg_check(const double *rate, const int32_t *next, size_t count, int *flag,
                               hipStream_t s);

// colpanel + main: applies the bt pivots to every row of the slab.
template <typename T>
hipError_t launch_fused_relax(const FusedArgs<T> &a, hipStream_t s, int skip_lo = 0, int skip_hi = 0);
// The two halves separately: pivot-column snapshots for all rows of the slab, then the main
// kernel on local rows [r_lo, r_hi) (used by the look-ahead schedule).
template <typename T> hipError_t launch_fused_colpanel(const FusedArgs<T> &a, hipStream_t s);
// Column controls of a main launch (symmetric look-ahead): c_hi > c_lo restricts the launch to the
// columns [c_lo, c_hi) (multiples of 64), skip_hi > skip_lo leaves the columns [skip_lo, skip_hi)
// (multiples of 4) to another launch of the same pass.  Default: all columns.
struct FusedCols {
    int c_lo = 0, c_hi = 0, skip_lo = 0, skip_hi = 0;
    static FusedCols only(int lo, int hi) { FusedCols c; c.c_lo = lo; c.c_hi = hi; return c; }
    static FusedCols except(int lo, int hi) { FusedCols c; c.skip_lo = lo; c.skip_hi = hi; return c; }
};
// True if the main launch these arguments select leaves no room for a panel workgroup when ONE of its own
// workgroups retires (f32 with next-hops on 64 x 64 tiles: four workgroups per CU at 128 registers and 36.5 KB of
// LDS each -- the hole one of them leaves is smaller than a panel workgroup's 48 KB, and before round 4's panel
// rewrite also than its 4 x 48 registers per SIMD): the panels of the look-ahead chain then wait for the
// launch's tail.  A double-pass schedule issues such a main launch as two halves -- the first one's tail lets the
// chain's first panel in, the second one's the other (fused_range in fwx_api.hip, profiles/r04_timeline_*).
template <typename T> bool fused_main_starves_panels(const FusedArgs<T> &a);
// ... unless launch_fused_panels has a form for these arguments that fits the hole (a.side set: f32 with next-hops,
// no trace, no hops -- column workgroups of 32 rows, 32.25 KB of LDS and 32 registers): then the main launch stays
// whole and the chain runs beside it.
template <typename T> bool fused_panels_fit_beside(const FusedArgs<T> &a);
template <typename T>
hipError_t launch_fused_main(const FusedArgs<T> &a, int r_lo, int r_hi, hipStream_t s,
                             int skip_lo = 0, int skip_hi = 0, FusedCols cols = FusedCols());

// rowpanel + colpanel of pass (a.k0, a.bt) in ONE launch, for a slab that is the whole matrix
// (a.rows == a.n, a.row0 == 0): the column panel evolves the diagonal block itself instead of
// reading a finished W, so neither waits for the other.  Writes w_out (and wh_out with hops), a.ct,
// a.cnt, a.cht and the trace's at_row / at_col; a.w / a.wh are not read.
template <typename T>
hipError_t launch_fused_panels(const FusedArgs<T> &a, T *w_out, int32_t *wh_out, hipStream_t s);

// diag + rowpanel: snapshot panel of the pivot rows `rows_base` (bt x n, at time k0); the matrix
// is not modified.  plog: the path trace AT THE SAME ROWS as rows_base (plog.last / plog.at_row
// point at pivot row k0; at_col is not touched).  hops_rows (same rows again) / wh: the hops of the
// pivot rows are carried through the panel and their time-k snapshots exported to wh (bt x n).
template <typename T>
hipError_t launch_fused_panel(const T *rows_base, int n, int k0, int bt, T *w, hipStream_t s,
                              PathLog plog = PathLog(), const int32_t *hops_rows = nullptr,
                              int32_t *wh = nullptr);

}  // namespace fwx
#endif
