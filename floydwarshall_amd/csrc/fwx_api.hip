// fwx_api.hip -- the C ABI of libfwx (include/fwx.h) over the gfx950 kernels.
//
// Replaces runAlgo (/root/reference/src/lib/Algorithms.hs:42-61) behind floydWarshall (:19-20).
// No CPU fallback: every solve entry point needs a HIP device and says so when there is none.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <new>

#include "fwx.h"
#include "fwx_guard.h"
#include "fwx_internal.h"
#include "fwx_kernels.h"
#include "fwx_replay.h"

using namespace fwxi;

namespace {

template <typename T>
int fused_block(const fwx_slab *sl, int k0, int bt, const T *w, const int32_t *wh,
                const fwx_fused_scratch *sc, const fwx_trace *tr, unsigned long long *d_updates,
                bool nonneg, hipStream_t s, int skip_lo = 0, int skip_hi = 0)
{
    fwx::FusedArgs<T> a;
    a.nonneg = nonneg;
    a.rate = (T *)sl->rate; a.next = sl->next; a.rows = sl->rows; a.n = sl->n; a.row0 = sl->row0;
    a.k0 = k0; a.bt = bt; a.w = w; a.ct = (T *)sc->col_rate; a.cnt = sc->col_next; a.updates = d_updates;
    a.ct_ld = (sl->rows + 3) & ~3;
    a.hops = sl->hops; a.wh = wh; a.cht = sc->col_hops;
    if (tr) { a.plog.last = tr->last; a.plog.at_col = tr->at_col; a.plog.at_row = tr->at_row; }
    hipError_t e = fwx::launch_fused_relax<T>(a, s, skip_lo, skip_hi);
    if (e == hipErrorInvalidValue) return FWX_ERR_INVALID;
    FWX_HIP(e);
    return FWX_OK;
}

// Can the fused kernels read this matrix?  (16-byte vectors along rows.)
template <typename T> bool fused_dims_ok(int n, const void *rate)
{
    return n % (16 / (int)sizeof(T)) == 0 && ((uintptr_t)rate % 16) == 0;
}
template <typename T> bool fused_ok(int n, const void *rate, const int32_t *)
{
    return fused_dims_ok<T>(n, rate);
}

// Single-GPU solve of pivots [k_begin,k_end) with the fused engine.  Per block of <= 64 pivots:
// snapshot panel W_b (rowpanel), pivot-column snapshots for all rows (colpanel), main kernel over
// all rows.  Three schedules:
//   serial: ONE fused_panels launch (row panel + column panel as one grid) and ONE main launch
//     per pass, in one stream.  While a main launch is not much longer than a panel kernel
//     (20-40 us) any look-ahead costs more -- an extra latency-bound launch and two cross-stream
//     event hops (6 + 12 us) -- than its overlap gives: N = 1024 f32 1.08 ms (rows-only look-ahead)
//     -> 0.73 (serial, three launches) -> 0.52 (panels merged); and because the panels then run on
//     an idle chip (25 us instead of 75 us beside a main launch) it stays level with the look-ahead
//     forms up to the largest sizes for rates-only solves.
//   symmetric look-ahead (large matrices with next-hops, 64-aligned blocks): the panel chain of
//     block b+1 only needs the 64 pivot ROWS and the 64 pivot COLUMNS of b+1 as pass b leaves them.
//     The side stream relaxes exactly those with pass b's panels, then runs fused_panels(b+1) --
//     all of it beside main(b), which leaves those rows and columns alone.  The main stream carries
//     nothing but main kernels back to back (gap 44 -> 12.5 us).
//   rows-only look-ahead (the fallback for pivot ranges that do not start on a multiple of 64):
//     main(b) on the next block's rows first, their rowpanel on the side stream while main(b)
//     sweeps the rest; colpanel(b+1) follows main(b).
// Crossovers (serial vs symmetric, ms; gpurun_out/r02_run37.log, r02_run38.log): f32 rates 8192:
// 21.6 / 22.3, 12288: 69.1 / 69.9, 16384: 156.3 / 157.0 -> always serial; f32 + next 8192: 39.5 / 39.6,
// 12288: 117.0 / 117.5, 16384: 266.3 / 262.2; + trace 8192: 46.1 / 45.8, 10240: 84.6 / 83.6, 16384:
// 316.2 / 305.0; f64 + trace 8192: 137.9 / 132.6.
// ws: see fused_ws_bytes.
//   double pass (inside the domain -- the max-form and the arg kernels --, 64-aligned blocks, large
//     matrices): the main kernel applies TWO passes (128 pivots) per launch, which halves the tile
//     traffic and the per-tile overhead that kept it at 0.75 of its issue bound (the arg kernels run
//     their two passes one after the other on the tile they keep in registers).  It needs the
//     panels of both passes up front, so the look-ahead is two deep: beside main(P) -- which leaves
//     the 128 rows and 128 columns of the NEXT pair of blocks alone -- the side stream brings
//     exactly that cross up to date with the pair being applied (two launches of 128 pivots), runs
//     the panels of the first block of the next pair, applies that one pass to the rows and columns
//     of the second block, and runs its panels.  An entry that main(P+1) then folds again with a
//     pass it has already seen does not move (max is idempotent; the compare form would not move or
//     count it either -- and an entry that does not move is no item of the arg re-scan: its next-hop,
//     path length and trace stay), so main(P+1) simply covers everything but the cross after it.
constexpr int kLookaheadMinN = INT32_MAX, kLookaheadMinNWithNext = 16384, kLookaheadMinNWithTrace = 8192;
// crossover (single / double pass, ms; profiles/r03_double_pass_crossover.txt): f32 4096: 4.10 / 4.74,
// 6144: 10.47 / 10.18, 8192: 21.2 / 20.0, 12288: 68.1 / 61.0, 16384: 152.0 / 135.3; f64 6144: 23.1 / 22.1,
// 16384: 355 / 324 -- below ~6000 the side chain (five launches per 128 pivots) is the critical path
constexpr int kDoublePassMinN = 6144;    // FWX_DOUBLE_PASS_MIN_N overrides
// ... and with next-hops (the arg kernels; + trace, + hops), f32 only: FWX_DOUBLE_PASS_NEXT_MIN_N overrides.
// (f64: the two-pass fused_main_arg_f64 is SLOWER than two launches, N = 16384 + next 486 -> 499 ms on
// one box -- tools/runs/r03_run33.sh --, so f64 stays on the single pass unless the variable asks)
// (round 4, after the panel flags and the 32-row column panels: tools/runs/r04_run42.sh, single / double pass, ms:
//  + next 4096 5.8 / 6.0, 5120 9.9 / 9.7, 6144 16.2 / 16.0, 7168 24.3 / 23.8; + trace 4096 6.3 / 6.8, 6144 21.0 /
//  17.9, 7168 32.0 / 26.1 -- the threshold was 8192)
constexpr int kDoublePassNextMinN = 5120;
// FWX_LOOKAHEAD_MIN_N / FWX_SYMMETRIC_MIN_N override the thresholds (tests force each schedule at
// small sizes, tuning runs switch one off with a huge value); read on every solve.
static int env_threshold(const char *name, int dflt)
{
    const char *e = getenv(name);
    if (e && *e) {
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end != e && v >= 0 && v <= INT32_MAX) return (int)v;
    }
    return dflt;
}

template <typename T>
int fused_range(T *rate, int32_t *next, int32_t *hops, int n, int k_begin, int k_end, void *ws,
                unsigned long long *d_updates, hipStream_t s, fwx::PathLog plog, bool nonneg,
                SideStream *kept_side = nullptr, Resume *rec = nullptr)
{
    char *p = (char *)ws;
    const int ld = (n + 3) & ~3;
    T *wbuf[2], *ctbuf[2];
    int32_t *cntbuf[2];
    int32_t *whbuf[2] = {nullptr, nullptr}, *chtbuf[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b) { wbuf[b] = (T *)p;          p += (size_t)FWX_FUSED_B * n * sizeof(T); }
    for (int b = 0; b < 2; ++b) { ctbuf[b] = (T *)p;         p += (size_t)FWX_FUSED_B * ld * sizeof(T); }
    for (int b = 0; b < 2; ++b) { cntbuf[b] = (int32_t *)p;  p += (size_t)FWX_FUSED_B * ld * sizeof(int32_t); }
    if (hops) {                    // hops panels of the pivot rows + hops of the pivot columns
        for (int b = 0; b < 2; ++b) { whbuf[b] = (int32_t *)p;  p += (size_t)FWX_FUSED_B * n * sizeof(int32_t); }
        for (int b = 0; b < 2; ++b) { chtbuf[b] = (int32_t *)p; p += (size_t)FWX_FUSED_B * ld * sizeof(int32_t); }
    }
    if (k_end <= k_begin) return FWX_OK;
    // Resumable handle: every pass writes its panels to their place in the all-pivot arrays (row k of
    // each = pivot k) instead of a ping-pong buffer, and the state is copied at the checkpoint pivots.
    if (rec && k_begin % FWX_FUSED_B) rec = nullptr;      // passes must start on multiples of 64
    auto panel_set = [&](int k0, int b) {
        if (!rec) return;
        wbuf[b] = (T *)rec->w + (size_t)k0 * n;
        ctbuf[b] = (T *)rec->ct + (size_t)k0 * ld;
        cntbuf[b] = rec->cnt ? rec->cnt + (size_t)k0 * ld : nullptr;
        whbuf[b] = rec->wh ? rec->wh + (size_t)k0 * n : nullptr;
        chtbuf[b] = rec->cht ? rec->cht + (size_t)k0 * ld : nullptr;
    };
    auto checkpoint = [&](int k0) -> int {        // the state at the START of step k0, if it is one
        if (!rec) return FWX_OK;
        for (size_t c = 0; c < rec->pivot.size(); ++c) {
            if (rec->pivot[c] != k0) continue;
            const size_t nn = (size_t)n * n;
            FWX_HIP(hipMemcpyAsync(rec->rate[c], rate, nn * sizeof(T), hipMemcpyDeviceToDevice, s));
            if (next) FWX_HIP(hipMemcpyAsync(rec->next[c], next, nn * 4, hipMemcpyDeviceToDevice, s));
            if (hops) FWX_HIP(hipMemcpyAsync(rec->hops[c], hops, nn * 4, hipMemcpyDeviceToDevice, s));
            if (plog.last) {
                FWX_HIP(hipMemcpyAsync(rec->last[c], plog.last, nn * 4, hipMemcpyDeviceToDevice, s));
                FWX_HIP(hipMemcpyAsync(rec->at_col[c], plog.at_col, nn * 4, hipMemcpyDeviceToDevice, s));
                FWX_HIP(hipMemcpyAsync(rec->at_row[c], plog.at_row, nn * 4, hipMemcpyDeviceToDevice, s));
            }
        }
        return FWX_OK;
    };
    SideStream local_side;
    SideStream &side = kept_side ? *kept_side : local_side;
    if (!side.s) {
        const int rc = side.init();
        if (rc) return rc;
    }
    Throttle thr;

    fwx::FusedArgs<T> a;
    a.rate = rate; a.next = next; a.rows = n; a.n = n; a.row0 = 0;
    a.ct_ld = ld; a.updates = d_updates;
    a.nonneg = nonneg;     // the caller has run the domain check (route_solve)
    a.plog = plog;         // path trace: kept by all three kernels of a pass (needs next)
    a.hops = hops;         // hops: carried by the panels, written by the main kernel (needs next)
    auto bind = [&](int k0, int bt, int bi) {       // pass (k0, bt) reads / writes buffer set bi
        panel_set(k0, bi);
        a.k0 = k0; a.bt = bt; a.w = wbuf[bi]; a.wh = whbuf[bi];
        a.ct = ctbuf[bi]; a.cnt = next ? cntbuf[bi] : nullptr; a.cht = chtbuf[bi];
    };
    auto rowpanel = [&](int k0, int bt, int bi, hipStream_t st) {
        panel_set(k0, bi);
        return fwx::launch_fused_panel<T>(rate + (size_t)k0 * n, n, k0, bt, wbuf[bi], st,
                                          plog_rows(plog, (size_t)k0 * n),
                                          hops ? hops + (size_t)k0 * n : nullptr, whbuf[bi]);
    };
    const bool lookahead = n >= env_threshold("FWX_LOOKAHEAD_MIN_N",
                                               !next ? kLookaheadMinN
                                                     : plog.last ? kLookaheadMinNWithTrace : kLookaheadMinNWithNext);
    const bool symmetric_ok = n >= env_threshold("FWX_SYMMETRIC_MIN_N", 0) && k_begin % FWX_FUSED_B == 0;

    auto panels = [&](int k0, int bt, int bi, hipStream_t st) {   // both panels of a pass, one launch
        bind(k0, bt, bi);
        return fwx::launch_fused_panels<T>(a, wbuf[bi], whbuf[bi], st);
    };

    // ---- double pass: see the header comment ----------------------------------------------------
    if (nonneg && !d_updates && !rec && k_begin % FWX_FUSED_B == 0 &&
        n >= (next ? env_threshold("FWX_DOUBLE_PASS_NEXT_MIN_N", sizeof(T) == 4 ? kDoublePassNextMinN : INT32_MAX)
                   : env_threshold("FWX_DOUBLE_PASS_MIN_N", kDoublePassMinN)) &&
        k_end - k_begin >= 4 * FWX_FUSED_B) {
        constexpr int Bq = FWX_FUSED_B;
        const int nb = (k_end - k_begin) / Bq;          // full blocks; a ragged tail is handled below
        const int pairs = nb / 2;
        // four panel sets, as two adjacent pairs: block q lives in set q & 3, so a pair (2P, 2P + 1)
        // is contiguous in W (128 rows), in Ct (128 lines) and likewise in CNt, WH, CHt
        char *q4 = (char *)ws;
        T *w4 = (T *)q4;                 q4 += (size_t)4 * Bq * n * sizeof(T);
        T *ct4 = (T *)q4;                q4 += (size_t)4 * Bq * ld * sizeof(T);
        int32_t *cnt4 = (int32_t *)q4;   q4 += (size_t)4 * Bq * ld * sizeof(int32_t);
        int32_t *wh4 = nullptr, *cht4 = nullptr;
        if (hops) {
            wh4 = (int32_t *)q4;         q4 += (size_t)4 * Bq * n * sizeof(int32_t);
            cht4 = (int32_t *)q4;
        }
        auto set_of = [&](int q) { return q & 3; };
        auto wh_of = [&](int q) { return wh4 ? wh4 + (size_t)set_of(q) * Bq * n : nullptr; };
        auto bind4 = [&](int q, int blocks) {           // pivots of `blocks` blocks starting at block q
            a.k0 = k_begin + q * Bq; a.bt = blocks * Bq;
            a.w = w4 + (size_t)set_of(q) * Bq * n; a.wh = wh_of(q);
            a.ct = ct4 + (size_t)set_of(q) * Bq * ld;
            a.cnt = next ? cnt4 + (size_t)set_of(q) * Bq * ld : nullptr;
            a.cht = cht4 ? cht4 + (size_t)set_of(q) * Bq * ld : nullptr;
        };
        auto panels4 = [&](int q, hipStream_t st) {
            bind4(q, 1);
            a.side = st != s;                       // beside a main launch: the panel kernels may pick a leaner form
            const hipError_t e = fwx::launch_fused_panels<T>(a, w4 + (size_t)set_of(q) * Bq * n, wh_of(q), st);
            a.side = false;
            return e;
        };
        // pivots of `blocks` blocks from block q onto the rows [lo, hi) (all columns) and the columns
        // [lo, hi) (the other rows)
        auto cross = [&](int q, int blocks, int lo, int hi, hipStream_t st) -> hipError_t {
            if (hi <= lo) return hipSuccess;
            bind4(q, blocks);
            a.side = true;
            hipError_t e = fwx::launch_fused_main<T>(a, lo, hi, st);
            if (e == hipSuccess) e = fwx::launch_fused_main<T>(a, 0, n, st, lo, hi, fwx::FusedCols::only(lo, hi));
            a.side = false;
            return e;
        };
        auto rows_of = [&](int q) { return k_begin + q * Bq; };
        // chain(0): panels of block 0, that pass onto block 1's rows and columns, panels of block 1
        FWX_HIP(panels4(0, s));
        FWX_HIP(cross(0, 1, rows_of(1), rows_of(2), s));
        FWX_HIP(panels4(1, s));
        for (int P = 0; P < pairs; ++P) {
            const int q = 2 * P;
            const int q_lo = q + 2, q_hi = q + 4 < nb ? q + 4 : nb;     // blocks of the next pair (or the odd last one)
            const int x_lo = rows_of(q_lo), x_hi = rows_of(q_hi);
            if (q_hi > q_lo) {
                FWX_HIP(hipEventRecord(side.main_done, s));       // main(P - 1) and chain(P) precede
                FWX_HIP(hipStreamWaitEvent(side.s, side.main_done, 0));
                FWX_HIP(cross(q, 2, x_lo, x_hi, side.s));         // the pair being applied onto the next cross
                FWX_HIP(panels4(q_lo, side.s));
                if (q_hi - q_lo == 2) {
                    FWX_HIP(cross(q_lo, 1, rows_of(q_lo + 1), x_hi, side.s));
                    FWX_HIP(panels4(q_lo + 1, side.s));
                }
                FWX_HIP(hipEventRecord(side.panel_done, side.s));
                bind4(q, 2);
                // a main launch whose retiring workgroups leave no room for a panel workgroup -- and for which the panel
                // kernels have no form that fits the hole (the path trace) -- goes out as two halves: the chain's
                // panels get in at each half's tail instead of at the end of the whole launch (N = 8192 + next-hops:
                // gap between main launches 77 -> 20 us; FWX_SPLIT_MAIN=0: A/B)
                static const bool split_ok = [] { const char *e = getenv("FWX_SPLIT_MAIN"); return !(e && *e == '0'); }();
                const int h = n / 2 / 128 * 128;
                if (split_ok && h > 0 && fwx::fused_main_starves_panels<T>(a) && !fwx::fused_panels_fit_beside<T>(a)) {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, h, s, x_lo, x_hi, fwx::FusedCols::except(x_lo, x_hi)));
                    FWX_HIP(fwx::launch_fused_main<T>(a, h, n, s, x_lo, x_hi, fwx::FusedCols::except(x_lo, x_hi)));
                } else {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s, x_lo, x_hi, fwx::FusedCols::except(x_lo, x_hi)));
                }
                FWX_HIP(hipStreamWaitEvent(s, side.panel_done, 0));
            } else {
                bind4(q, 2);
                FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s));
            }
            const int rc = thr.tick(s, 8);
            if (rc) return rc;
        }
        if (nb & 1) {                                   // the odd last block: its panels are ready
            bind4(nb - 1, 1);
            FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s));
        }
        FWX_HIP(hipStreamSynchronize(s));
        k_begin += nb * Bq;                             // a ragged tail (< 64 pivots) takes the serial form
        if (k_end <= k_begin) return FWX_OK;
        bind(k_begin, k_end - k_begin, 0);              // (restores the two-set bindings below)
    }

    int bi = 0;
    bool col_ready = false;        // colpanel of the current pass already ran
    {
        const int bt = k_end - k_begin < FWX_FUSED_B ? k_end - k_begin : FWX_FUSED_B;
        if (!lookahead) {
            FWX_HIP(panels(k_begin, bt, 0, s));
            col_ready = true;
        } else {
            FWX_HIP(rowpanel(k_begin, bt, 0, s));
        }
    }
    for (int k0 = k_begin; k0 < k_end; k0 += FWX_FUSED_B, bi ^= 1) {
        const int bt = k_end - k0 < FWX_FUSED_B ? k_end - k0 : FWX_FUSED_B;
        const int k1 = k0 + bt;
        // everything queued on `s` so far (the previous pass, whose side chain `s` has waited for)
        // precedes this copy; this pass's panels only READ the matrix
        if (k0 > k_begin) { const int rc = checkpoint(k0); if (rc) return rc; }
        bind(k0, bt, bi);
        if (!col_ready) FWX_HIP(fwx::launch_fused_colpanel<T>(a, s));
        col_ready = false;
        if (k1 < k_end && !lookahead) {
            const int bt1 = k_end - k1 < FWX_FUSED_B ? k_end - k1 : FWX_FUSED_B;
            FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s));
            FWX_HIP(panels(k1, bt1, bi ^ 1, s));
            col_ready = true;
        } else if (k1 < k_end) {
            const int bt1 = k_end - k1 < FWX_FUSED_B ? k_end - k1 : FWX_FUSED_B;
            if (symmetric_ok && bt1 == FWX_FUSED_B) {
                // everything up to here (the previous main, this pass's panels) precedes the side chain
                FWX_HIP(hipEventRecord(side.main_done, s));
                FWX_HIP(hipStreamWaitEvent(side.s, side.main_done, 0));
                // side: pass b on the next block's rows (all columns), then on its columns (the
                // other rows) ...
                a.side = true;
                hipError_t se = fwx::launch_fused_main<T>(a, k1, k1 + bt1, side.s);
                if (se == hipSuccess)
                    se = fwx::launch_fused_main<T>(a, 0, n, side.s, k1, k1 + bt1, fwx::FusedCols::only(k1, k1 + bt1));
                a.side = false;
                FWX_HIP(se);
                // ... then the next pass's panels (one launch) into the other buffer set
                FWX_HIP(panels(k1, bt1, bi ^ 1, side.s));
                FWX_HIP(hipEventRecord(side.panel_done, side.s));
                // main: pass b everywhere else
                bind(k0, bt, bi);
                FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s, k1, k1 + bt1,
                                                  fwx::FusedCols::except(k1, k1 + bt1)));
                FWX_HIP(hipStreamWaitEvent(s, side.panel_done, 0));
                col_ready = true;
            } else {
                // the next panel's rows first ...
                FWX_HIP(fwx::launch_fused_main<T>(a, k1, k1 + bt1, s));
                FWX_HIP(hipEventRecord(side.rows_done, s));
                FWX_HIP(hipStreamWaitEvent(side.s, side.rows_done, 0));
                // ... their snapshot panel on the side stream ...
                FWX_HIP(rowpanel(k1, bt1, bi ^ 1, side.s));
                FWX_HIP(hipEventRecord(side.panel_done, side.s));
                // ... while the rest of the matrix is relaxed on the main stream
                if (k1 % 8 == 0 && bt1 % 8 == 0) {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s, k1, k1 + bt1));   // one launch, rows skipped
                } else {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, k1, s));
                    FWX_HIP(fwx::launch_fused_main<T>(a, k1 + bt1, n, s));
                }
                FWX_HIP(hipStreamWaitEvent(s, side.panel_done, 0));
            }
        } else {
            FWX_HIP(fwx::launch_fused_main<T>(a, 0, n, s));
        }
        const int rc = thr.tick(s, 6);
        if (rc) return rc;
    }
    // the side stream's work is ordered before `s` by the last panel_done wait (or it never ran)
    FWX_HIP(hipStreamSynchronize(s));
    return FWX_OK;
}

// AUTO: the single-launch kernel up to its 64 x 64 form, the fused engine above wherever it applies.
constexpr int kSmallSolveAutoMax = 64;
template <typename T> bool pick_fused(int engine, int n, int nd, const void *rate, const int32_t *hops)
{
    if (engine == FWX_ENGINE_PERK) return false;
    if (!fused_ok<T>(nd, rate, hops)) return false;
    // tools/measure_small.py (profiles/r02_small_sizes.txt): with the serial two-launch schedule
    // the fused engine beats the per-k engine at every order, and the single-launch kernel above
    // its 64 x 64 register form (n = 72 f64 + next: 0.13 ms against 0.20; n = 128: 0.18 / 0.36)
    return engine == FWX_ENGINE_FUSED || n > kSmallSolveAutoMax;
}

// Which engine runs a solve of pivots [k_begin, k_end) of a matrix of order n held on the device at
// order (= pitch) nd >= n: the host-buffer path and the handles pad an odd order with inert entries
// (fwx_matrix::nd), every engine then runs the nd x nd arrays over the real pivots only.
enum Route { ROUTE_SMALL, ROUTE_PERK, ROUTE_FUSED };
template <typename T>
int route_solve(const Opts &op, int n, int nd, const T *rate, const int32_t *next,
                const int32_t *hops, bool counting, int *d_flag, hipStream_t s, Route &route,
                bool &nonneg, fwx_matrix *cache = nullptr)
{
    nonneg = false;
    if (op.engine == FWX_ENGINE_FUSED && !fused_ok<T>(nd, rate, hops)) return FWX_ERR_UNSUPPORTED;
    // AUTO below the fused engine's range, or where it cannot read the matrix / the input lies outside
    // its domain: the single-launch kernel while it fits (nd <= FWX_SMALL_N), else one launch per pivot
    const Route fallback = (op.engine == FWX_ENGINE_AUTO && nd <= FWX_SMALL_N) ? ROUTE_SMALL : ROUTE_PERK;
    if (op.engine == FWX_ENGINE_AUTO && n <= kSmallSolveAutoMax) { route = ROUTE_SMALL; return FWX_OK; }
    if (!pick_fused<T>(op.engine, n, nd, rate, hops)) { route = fallback; return FWX_OK; }
    // The fused kernels take next[i][k] as the head of ikPath ++ kjPath (Algorithms.hs:55), which
    // is the reference's list head only while a winning product never has an empty ikPath -- true
    // on the reference's own domain, checked here.  Outside it the per-k engine, which reads the
    // pivot row's next-hops for exactly that case, runs the solve (same bits as the reference).
    int bits = 0;
    if (next || !counting) {
        // a handle remembers the answer for its arrays (fwx_matrix::dom_known); with next-hops the
        // check reads both arrays, without them only bit 0 is meaningful -- a cached answer taken
        // with next-hops serves both
        if (cache && cache->dom_known) {
            bits = cache->dom_bits;
        } else {
            const int rc = domain_bits<T>(rate, next, (size_t)nd * nd, d_flag, s, bits);
            if (rc) return rc;
            if (cache && (next || !cache->next)) { cache->dom_bits = bits; cache->dom_known = 1; }
        }
    }
    if (next && bits != 3) { route = fallback; return FWX_OK; }
    nonneg = !counting && (next ? bits == 3 : (bits & 1) != 0);   // max-form kernels allowed
    route = ROUTE_FUSED;
    return FWX_OK;
}

template <typename T>
int solve_host(int32_t n, T *rate, int32_t *next, int32_t *hops, const fwx_opts *o)
{
    if (n < 0) return FWX_ERR_INVALID;
    if (n == 0) return FWX_OK;  // empty map -> empty matrix (AlgorithmsTest.hs:62-64)
    if (!rate) return FWX_ERR_INVALID;
    if (hops && !next) return FWX_ERR_INVALID;
    Opts op;
    int rc = read_opts(o, n, op);
    if (rc) return rc;
    DeviceGuard g;
    if ((rc = g.enter(op.device))) return rc;

    // Device order of the matrix.  The fused engine needs rows that are a multiple of 16 bytes; an
    // odd-sized matrix headed for it is PADDED on the device with +0.0 rows and columns.  Padding
    // never reaches a real entry -- step k reads r[i][k] and r[k][j] of real pivots k only, and a
    // padding entry is never a pivot row or column -- and a +0.0 target is never improved
    // (0 < +-0 and 0 < NaN are false), so U is unchanged too.  The caller's arrays stay n x n.
    constexpr int VW = 16 / (int)sizeof(T);
    const bool to_fused = op.engine == FWX_ENGINE_FUSED || (op.engine == FWX_ENGINE_AUTO && n > kSmallSolveAutoMax);
    const int nd = (to_fused && n % VW) ? (n + VW - 1) / VW * VW : n;
    const size_t nn = (size_t)nd * (size_t)nd;
    // stream, look-ahead stream, device buffers, workspace: a pooled per-call context (CallCtx)
    CtxLease lease;
    if ((rc = lease.open())) return rc;
    CallCtx &cx = *lease.c;
    if (op.has_stream) cx.uses_stream(op.stream);   // drained before the context's buffers are reused
    struct { void *p = nullptr; } d_rate, d_next, d_hops, d_upd;
    if ((rc = cx.reserve(CallCtx::RATE, nn * sizeof(T), &d_rate.p))) return rc;
    if (next && (rc = cx.reserve(CallCtx::NEXT, nn * sizeof(int32_t), &d_next.p))) return rc;
    if (hops && (rc = cx.reserve(CallCtx::HOPS, nn * sizeof(int32_t), &d_hops.p))) return rc;
    if ((rc = cx.reserve(CallCtx::SMALL, (FWX_UPDATE_SHARDS + 2) * sizeof(unsigned long long), &d_upd.p))) return rc;
    int *d_flag = (int *)((unsigned long long *)d_upd.p + FWX_UPDATE_SHARDS);

    hipStream_t s = op.has_stream ? op.stream : cx.s;   // a non-blocking stream, never the null stream
    auto copy2d = [&](void *dst, size_t dpitch, const void *src, size_t spitch, size_t es,
                      hipMemcpyKind kind) -> int {
        if (dpitch == spitch)
            FWX_HIP(hipMemcpyAsync(dst, src, (size_t)n * (size_t)n * es, kind, s));
        else
            FWX_HIP(hipMemcpy2DAsync(dst, dpitch * es, src, spitch * es, (size_t)n * es, (size_t)n, kind, s));
        return FWX_OK;
    };
    if (nd != n) {
        FWX_HIP(hipMemsetAsync(d_rate.p, 0, nn * sizeof(T), s));                  // +0.0
        if (next) FWX_HIP(hipMemsetAsync(d_next.p, 0xFF, nn * sizeof(int32_t), s));   // -1
        if (hops) FWX_HIP(hipMemsetAsync(d_hops.p, 0, nn * sizeof(int32_t), s));
    }
    // Small matrices travel through the context's pinned staging buffer and a plain memcpy: an
    // asynchronous copy from / to pageable memory makes the runtime pin and unpin the caller's pages
    // per call, which costs more than the transfer itself for small arrays (fwx_solve_f64 with next +
    // hops, n = 4: 0.079 -> 0.048 ms, n = 64: 0.155 -> 0.123, n = 256: 1.04 -> 0.92); larger ones go
    // straight from / to the caller's arrays (CallCtx::kStageBytes).
    const size_t b_rate = (size_t)n * n * sizeof(T), b_idx = (size_t)n * n * sizeof(int32_t);
    const size_t b_total = b_rate + (next ? b_idx : 0) + (hops ? b_idx : 0);
    const bool staged = b_total <= CallCtx::kStageBytes;
    char *st_rate = nullptr, *st_next = nullptr, *st_hops = nullptr;
    if (staged) {
        void *pinned = nullptr;
        if ((rc = cx.reserve_pinned(b_total, &pinned))) return rc;
        st_rate = (char *)pinned;
        st_next = st_rate + b_rate;
        st_hops = st_next + (next ? b_idx : 0);
        memcpy(st_rate, rate, b_rate);
        if (next) memcpy(st_next, next, b_idx);
        if (hops) memcpy(st_hops, hops, b_idx);
    }
    if ((rc = copy2d(d_rate.p, nd, staged ? (const void *)st_rate : (const void *)rate, n, sizeof(T), hipMemcpyHostToDevice))) return rc;
    if (next && (rc = copy2d(d_next.p, nd, staged ? (const void *)st_next : (const void *)next, n, sizeof(int32_t), hipMemcpyHostToDevice))) return rc;
    if (hops && (rc = copy2d(d_hops.p, nd, staged ? (const void *)st_hops : (const void *)hops, n, sizeof(int32_t), hipMemcpyHostToDevice))) return rc;
    FWX_HIP(hipMemsetAsync(d_upd.p, 0, FWX_UPDATE_SHARDS * sizeof(unsigned long long), s));

    T *dr = (T *)d_rate.p;
    int32_t *dn = (int32_t *)d_next.p, *dh = (int32_t *)d_hops.p;
    unsigned long long *upd = op.updates_out ? (unsigned long long *)d_upd.p : nullptr;
    Route route;
    bool nonneg = false;
    // (a padded matrix is routed by its real order n and solved over its real pivots only: the padding
    // is inert)
    if ((rc = route_solve<T>(op, n, nd, dr, dn, dh, upd != nullptr, d_flag, s, route, nonneg))) return rc;
    if (route == ROUTE_SMALL) {
        // the reference's own regime: the whole solve in one single-workgroup launch
        FWX_HIP(fwx::launch_small_solve<T>(dr, dn, dh, nd, op.k_begin, op.k_end, upd, fwx::PathLog(), s));
    } else if (route == ROUTE_FUSED) {
        void *ws = nullptr;
        if ((rc = cx.reserve(CallCtx::WS, fused_ws_bytes(nd, sizeof(T), dh != nullptr), &ws))) return rc;
        rc = fused_range<T>(dr, dn, dh, nd, op.k_begin, op.k_end, ws, upd, s, fwx::PathLog(), nonneg, &cx.side);
        if (rc) return rc;
    } else {
        // per-k engine; pitch nd (a matrix padded for the fused engine but found outside its domain)
        rc = relax_range<T>(dr, dn, dh, nd, nd, 0, dr + (size_t)op.k_begin * nd,
                            dh ? dh + (size_t)op.k_begin * nd : nullptr, nd, op.k_begin, op.k_end,
                            op.serpentine, upd, s, fwx::PathLog(), 0, 0,
                            dn ? dn + (size_t)op.k_begin * nd : nullptr);
        if (rc) return rc;
    }

    if ((rc = copy2d(staged ? (void *)st_rate : (void *)rate, n, d_rate.p, nd, sizeof(T), hipMemcpyDeviceToHost))) return rc;
    if (next && (rc = copy2d(staged ? (void *)st_next : (void *)next, n, d_next.p, nd, sizeof(int32_t), hipMemcpyDeviceToHost))) return rc;
    if (hops && (rc = copy2d(staged ? (void *)st_hops : (void *)hops, n, d_hops.p, nd, sizeof(int32_t), hipMemcpyDeviceToHost))) return rc;
    FWX_HIP(hipStreamSynchronize(s));
    if (staged) {
        memcpy(rate, st_rate, b_rate);
        if (next) memcpy(next, st_next, b_idx);
        if (hops) memcpy(hops, st_hops, b_idx);
    }
    if (op.updates_out) {
        if ((rc = sum_updates((unsigned long long *)d_upd.p, op.updates_out, s))) return rc;
    }
    return FWX_OK;
}

// Single-thread device walk of the next-hop matrix (fwx_matrix_query).
__global__ void follow_path_kernel(const int32_t *next, int n, int ld, int src, int dst, int32_t *out,
                                   int cap, int32_t *len_out)
{
    int len = 0, cur = src;
    if (next[(size_t)src * ld + dst] < 0) { *len_out = 0; return; }
    while (cur != dst || len == 0) {
        const int nx = next[(size_t)cur * ld + dst];
        if (nx < 0 || nx >= n || len >= n) { *len_out = FWX_ERR_CYCLE; return; }
        if (len >= cap) { *len_out = FWX_ERR_CAPACITY; return; }
        out[len++] = nx;
        cur = nx;
    }
    *len_out = len;
}

template <typename T>
int matrix_solve_typed(fwx_matrix *m, const Opts &op, unsigned long long *upd, hipStream_t s,
                       CallCtx *cx = nullptr);

// One thread per (src, dst) pair: batch path reconstruction from the next-hop matrix.
template <typename T>
__global__ __launch_bounds__(256) void follow_paths_kernel(const int32_t *next, int n, int count,
                                                           const int32_t *src, const int32_t *dst,
                                                           int32_t *len_out, const T *edge,
                                                           double *prod_out, int32_t *path_out,
                                                           int cap)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    const int s = src[q], d = dst[q];
    int len = 0, cur = s;
    double prod = 1.0;
    if (s < 0 || d < 0 || s >= n || d >= n) {
        len_out[q] = FWX_ERR_INVALID;
        return;
    }
    while (true) {
        const int nx = next[(size_t)cur * n + d];
        if (nx < 0) {
            if (len) len = FWX_ERR_CYCLE;    // a broken walk: only possible on inconsistent input
            break;
        }
        if (nx >= n || len >= n) { len = FWX_ERR_CYCLE; break; }
        if (edge) prod *= (double)edge[(size_t)cur * n + nx];
        if (path_out && len < cap) path_out[(size_t)q * cap + len] = nx;
        ++len;
        cur = nx;
        if (cur == d) break;
    }
    len_out[q] = len;
    if (prod_out) prod_out[q] = len > 0 ? prod : 0.0;
}

int check_slab(const fwx_slab *s)
{
    if (!s || s->n < 0 || s->rows < 0 || s->row0 < 0 || (int64_t)s->row0 + s->rows > s->n ||
        (s->dtype != FWX_F32 && s->dtype != FWX_F64) || (s->hops && !s->next))
        return FWX_ERR_INVALID;
    if (s->rows > 0 && s->n > 0 && !s->rate) return FWX_ERR_INVALID;
    return FWX_OK;
}

template <typename T>
int panel_impl(const fwx_slab *b, T *w, int32_t *w_hops, unsigned long long *d_updates,
                      hipStream_t s)
{
    const int n = b->n, k0 = b->row0, B = b->rows;
    T *rows = (T *)b->rate;
    fwx::RelaxArgs<T> a;
    a.rate = rows; a.next = b->next; a.hops = b->hops;
    a.rows = B; a.n = n; a.row0 = k0; a.updates = d_updates; a.flip = 0; a.plog = fwx::PathLog();
    for (int t = 0; t < B; ++t) {
        // Time-k snapshot of pivot row k = k0+t: every pivot < k has been applied, pivot k
        // leaves row k unchanged (Algorithms.hs:50), later pivots will change it.
        FWX_HIP(fwx::launch_snapshot_row<T>(w + (size_t)t * n, rows + (size_t)t * n,
                                            b->hops ? w_hops + (size_t)t * n : nullptr,
                                            b->hops ? b->hops + (size_t)t * n : nullptr, n, s));
        a.prow = w + (size_t)t * n;
        a.phops = b->hops ? w_hops + (size_t)t * n : nullptr;
        a.k = k0 + t;
        FWX_HIP(fwx::launch_relax<T>(a, s));
    }
    return FWX_OK;
}

}  // namespace

namespace {
template <typename T>
int matrix_solve_typed(fwx_matrix *m, const Opts &op, unsigned long long *upd, hipStream_t s, CallCtx *cx)
{
    const int n = m->n, nd = m->nd;      // order of the matrix; order (= pitch) of the device arrays
    T *r = (T *)m->rate;
    Route route;
    bool nonneg = false;
    int *d_flag = m->flag;
    int rc;
    if (!d_flag) {                       // fwx_dev_solve: a view of caller-owned memory + a pooled context
        if (!cx) return FWX_ERR_INVALID;
        void *small = nullptr;
        if ((rc = cx->reserve(CallCtx::SMALL, (FWX_UPDATE_SHARDS + 2) * sizeof(unsigned long long), &small))) return rc;
        d_flag = (int *)((unsigned long long *)small + FWX_UPDATE_SHARDS);
    }
    m->fresh = 0;                        // whatever happens next, the arrays are no longer the upload
    if ((rc = route_solve<T>(op, n, nd, r, m->next, m->hops, upd != nullptr, d_flag, s, route, nonneg,
                             m->flag ? m : nullptr)))
        return rc;
    if (route == ROUTE_SMALL) {
        if (m->resume) { m->resume->valid_upto = 0; m->resume->state_at = -1; }
        FWX_HIP(fwx::launch_small_solve<T>(r, m->next, m->hops, nd, op.k_begin, op.k_end, upd,
                                           m->plog, s));
        return FWX_OK;
    }
    if (route == ROUTE_FUSED) {
        // a handle keeps its workspace and look-ahead stream across solves; a view borrows the
        // pooled context's
        const size_t need = fused_ws_bytes(nd, sizeof(T), m->hops != nullptr);
        void *ws = nullptr;
        SideStream *side = nullptr;
        if (m->flag) {
            if (m->ws_bytes < need) {
                if (m->ws) {
                    drain_stream(s);
                    if (m->side) m->side->drain();
                    (void)hipFree(m->ws); m->ws = nullptr; m->ws_bytes = 0;
                }
                FWX_HIP(hipMalloc(&m->ws, need));
                m->ws_bytes = need;
            }
            if (!m->side) {
                m->side = new (std::nothrow) SideStream();
                if (!m->side) return FWX_ERR_OOM;
            }
            ws = m->ws;
            side = m->side;
        } else {
            if ((rc = cx->reserve(CallCtx::WS, need, &ws))) return rc;
            side = &cx->side;
        }
        // a resumable handle records the panels of every pass and the checkpoints it walks over; what
        // it holds beyond k_begin belongs to an older solve until this one has finished
        // a resumable handle records the panels of every pass and the checkpoints it walks over -- if
        // this solve continues the kept input's own solve (the arrays are that input at time k_begin)
        Resume *rec = nullptr;
        if (m->flag && m->resume) {
            Resume &R = *m->resume;
            const bool chain = m->kept_valid && R.state_at == op.k_begin && op.k_begin % FWX_FUSED_B == 0 &&
                               op.k_begin <= R.valid_upto;
            R.valid_upto = chain ? op.k_begin : 0;      // beyond k_begin: an older solve's, until this one ends
            R.state_at = -1;
            if (chain) rec = &R;
        }
        rc = fused_range<T>(r, m->next, m->hops, nd, op.k_begin, op.k_end, ws, upd, s, m->plog, nonneg, side, rec);
        if (rc) return rc;
        FWX_HIP(hipStreamSynchronize(s));
        if (rec) rec->valid_upto = rec->state_at = op.k_end;
        return FWX_OK;
    }
    if (m->resume) { m->resume->valid_upto = 0; m->resume->state_at = -1; }   // nothing was recorded
    return relax_range<T>(r, m->next, m->hops, nd, nd, 0, r + (size_t)op.k_begin * nd,
                          m->hops ? m->hops + (size_t)op.k_begin * nd : nullptr, nd, op.k_begin,
                          op.k_end, op.serpentine, upd, s, m->plog, 0, 0,
                          m->next ? m->next + (size_t)op.k_begin * nd : nullptr);
}

// Solve with the path trace (PathLog): one pass.  `last` starts at -1 everywhere; the kernels set
// it on every successful relaxation and copy its column k / row k into at_col / at_row at step k.
// resumed: the arrays AND the trace hold a restored checkpoint at time op.k_begin
// (fwx_matrix_resolve); the solve continues from there to the end.
int logged_solve(fwx_matrix *m, const Opts &op_in, hipStream_t s, bool resumed = false)
{
    if ((!resumed && op_in.k_begin != 0) || op_in.k_end != m->n)
        return FWX_ERR_UNSUPPORTED;              // the trace covers whole solves
    if (!resumed && !m->fresh) return FWX_ERR_INVALID;   // a traced solve starts from an uploaded input
    const Opts &op = op_in;                      // engines as for any matrix: single launch, fused
                                                 //   (no hops), per-k -- all three keep the trace
    const size_t nn = (size_t)m->nd * (size_t)m->nd;
    if (!resumed) {
        FWX_HIP(hipMemsetAsync(m->plog.last, 0xFF, nn * 4, s));
        FWX_HIP(hipMemsetAsync(m->plog.at_col, 0xFF, nn * 4, s));
        FWX_HIP(hipMemsetAsync(m->plog.at_row, 0xFF, nn * 4, s));
    }
    unsigned long long *upd = op.updates_out ? m->upd : nullptr;   // counting costs registers
    if (upd) FWX_HIP(hipMemsetAsync(upd, 0, FWX_UPDATE_SHARDS * 8, s));
    m->fresh = 0;
    m->last_u = 0;
    const int rc = m->dtype == FWX_F64 ? matrix_solve_typed<double>(m, op, upd, s)
                                       : matrix_solve_typed<float>(m, op, upd, s);
    if (rc) return rc;
    if (upd) {
        uint64_t u = 0;
        const int rc2 = sum_updates(upd, &u, s);   // synchronises
        if (rc2) return rc2;
        m->last_u = u;
        *op.updates_out = u;
    } else {
        FWX_HIP(hipStreamSynchronize(s));
    }
    m->rec_ready = 1;
    return FWX_OK;
}

// The reference's `_path` list of entry (src,dst), rebuilt from the path trace exactly as
// Algorithms.hs:55 built it: the newest update of (a,b) before time T, by pivot q, splits the path
// into path_q(a,q) ++ path_q(q,b); an entry with no update before T still has its buildMatrix path
// ([b] if next0[a][b] >= 0, else []).  T is the end of the solve for the query itself (`last`), and
// for every sub-entry it is the step named by one of its own indices: (a,q) at time q is read from
// at_col, (q,b) at time q from at_row.  Iterative, one thread; stack and output live in `walk`.
__global__ void exact_path_kernel(fwx::PathLog plog, const int32_t *next0, int n, int src, int dst,
                                  int32_t *walk, int cap, int32_t *len_out)
{
    enum { FINAL = 0, AS_COLUMN = 1, AS_ROW = 2 };
    int32_t *out = walk;                 // cap entries
    int32_t *stack = walk + cap;         // 3 * cap entries: (a, b, kind) triples
    int sp = 0, len = 0;
    stack[0] = src; stack[1] = dst; stack[2] = FINAL; sp = 1;
    while (sp > 0) {
        --sp;
        const int a = stack[3 * sp], b = stack[3 * sp + 1], kind = stack[3 * sp + 2];
        const size_t off = (size_t)a * n + b;
        const int q = kind == FINAL ? plog.last[off] : kind == AS_COLUMN ? plog.at_col[off] : plog.at_row[off];
        if (q < 0) {
            if (next0[off] >= 0) {
                if (len >= cap) { *len_out = FWX_ERR_CAPACITY; return; }
                out[len++] = b;
            }
        } else {
            if (sp + 2 > cap) { *len_out = FWX_ERR_CAPACITY; return; }
            stack[3 * sp] = q; stack[3 * sp + 1] = b; stack[3 * sp + 2] = AS_ROW; ++sp;      // second half
            stack[3 * sp] = a; stack[3 * sp + 1] = q; stack[3 * sp + 2] = AS_COLUMN; ++sp;   // first half
        }
    }
    *len_out = len;
}

// Batch form: one thread per (src[q], dst[q]); query q writes its list to paths + q*cap and uses
// stacks + q*3*cap as its stack.  len_out[q] = length, FWX_ERR_CAPACITY if it does not fit.
__global__ __launch_bounds__(64) void exact_paths_kernel(fwx::PathLog plog, const int32_t *next0, int n_real,
                                                         int n, int count, const int32_t *src,
                                                         const int32_t *dst, int32_t *paths,
                                                         int32_t *stacks, int cap, int32_t *len_out)
{
    enum { FINAL = 0, AS_COLUMN = 1, AS_ROW = 2 };
    const int qi = blockIdx.x * 64 + threadIdx.x;
    if (qi >= count) return;
    const int s0 = src[qi], d0 = dst[qi];
    if (s0 < 0 || d0 < 0 || s0 >= n_real || d0 >= n_real) { len_out[qi] = FWX_ERR_INVALID; return; }
    int32_t *out = paths + (size_t)qi * cap;
    int32_t *stack = stacks + (size_t)qi * 3 * cap;
    int sp = 0, len = 0;
    stack[0] = s0; stack[1] = d0; stack[2] = FINAL; sp = 1;
    while (sp > 0) {
        --sp;
        const int a = stack[3 * sp], b = stack[3 * sp + 1], kind = stack[3 * sp + 2];
        const size_t off = (size_t)a * n + b;                 // n: the pitch of the trace arrays
        const int q = kind == FINAL ? plog.last[off] : kind == AS_COLUMN ? plog.at_col[off] : plog.at_row[off];
        if (q < 0) {
            if (next0[off] >= 0) {
                if (len >= cap) { len_out[qi] = FWX_ERR_CAPACITY; return; }
                out[len++] = b;
            }
        } else {
            if (sp + 2 > cap) { len_out[qi] = FWX_ERR_CAPACITY; return; }
            stack[3 * sp] = q; stack[3 * sp + 1] = b; stack[3 * sp + 2] = AS_ROW; ++sp;
            stack[3 * sp] = a; stack[3 * sp + 1] = q; stack[3 * sp + 2] = AS_COLUMN; ++sp;
        }
    }
    len_out[qi] = len;
}
// The caller's arrays are n x n; a single-device handle holds them at pitch nd (fwx_matrix::nd).
// src / dst may be host or device memory (hipMemcpyDefault).
int copy_in(fwx_matrix *m, void *dev, const void *src, size_t es, hipStream_t s)
{
    const size_t n = (size_t)m->n, nd = (size_t)m->nd;
    if (nd == n) FWX_HIP(hipMemcpyAsync(dev, src, n * n * es, hipMemcpyDefault, s));
    else FWX_HIP(hipMemcpy2DAsync(dev, nd * es, src, n * es, n * es, n, hipMemcpyDefault, s));
    return FWX_OK;
}
int copy_out(fwx_matrix *m, void *dst, const void *dev, size_t es, hipStream_t s)
{
    const size_t n = (size_t)m->n, nd = (size_t)m->nd;
    if (nd == n) FWX_HIP(hipMemcpyAsync(dst, dev, n * n * es, hipMemcpyDefault, s));
    else FWX_HIP(hipMemcpy2DAsync(dst, n * es, dev, nd * es, n * es, n, hipMemcpyDefault, s));
    return FWX_OK;
}
// entry index of the caller's n x n view (i * n + j) -> offset in the device arrays
inline size_t dev_offset(const fwx_matrix *m, int64_t index)
{
    return (size_t)(index / m->n) * (size_t)m->nd + (size_t)(index % m->n);
}

// Element counts behind the memory a resumable handle keeps: `cells` per n x n array and `col_cells`
// per all-pivot column-panel array, summed over the partitions of a partitioned handle.
struct MultiDims { uint64_t cells, col_cells, w_cells; };
MultiDims resume_dims(const fwx_matrix *m)
{
    MultiDims d;
    if (m->multi) {
        multi_resume_dims(m, &d.cells, &d.col_cells, &d.w_cells);
        return d;
    }
    const uint64_t nd = (uint64_t)m->nd;
    d.cells = d.w_cells = nd * nd;
    d.col_cells = nd * ((nd + 3) & ~(uint64_t)3);
    return d;
}

void resume_free(Resume *r)
{
    if (!r) return;
    auto drop = [](void *p) { if (p) (void)hipFree(p); };
    for (void *p : r->rate) drop(p);
    for (auto *v : {&r->next, &r->hops, &r->last, &r->at_col, &r->at_row})
        for (int32_t *p : *v) drop(p);
    drop(r->w); drop(r->ct); drop(r->cnt); drop(r->wh); drop(r->cht); drop(r->idx);
    delete r;
}

// index: entry offsets in the DEVICE arrays (row * nd + column)
template <typename T>
int resolve_typed(fwx_matrix *m, int32_t count, const int64_t *index, int c_idx, hipStream_t s)
{
    Resume &R = *m->resume;
    const int n = m->nd, c = R.pivot[(size_t)c_idx];
    const size_t nn = (size_t)n * n;
    // the state at the start of step c ...
    FWX_HIP(hipMemcpyAsync(m->rate, R.rate[(size_t)c_idx], nn * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (m->next) FWX_HIP(hipMemcpyAsync(m->next, R.next[(size_t)c_idx], nn * 4, hipMemcpyDeviceToDevice, s));
    if (m->hops) FWX_HIP(hipMemcpyAsync(m->hops, R.hops[(size_t)c_idx], nn * 4, hipMemcpyDeviceToDevice, s));
    if (m->plog.last) {
        FWX_HIP(hipMemcpyAsync(m->plog.last, R.last[(size_t)c_idx], nn * 4, hipMemcpyDeviceToDevice, s));
        FWX_HIP(hipMemcpyAsync(m->plog.at_col, R.at_col[(size_t)c_idx], nn * 4, hipMemcpyDeviceToDevice, s));
        FWX_HIP(hipMemcpyAsync(m->plog.at_row, R.at_row[(size_t)c_idx], nn * 4, hipMemcpyDeviceToDevice, s));
    }
    // ... except the patched entries, replayed from the patched input through the stored panels into
    // the live arrays and into every checkpoint up to c (which thereby stay valid for the new input)
    ReplayTargets tg;
    memset(&tg, 0, sizeof(tg));
    for (int q = 0; q <= c_idx; ++q) {
        const int t = tg.count++;
        tg.pivot[t] = R.pivot[(size_t)q];
        tg.rate[t] = R.rate[(size_t)q];
        tg.next[t] = m->next ? R.next[(size_t)q] : nullptr;
        tg.hops[t] = m->hops ? R.hops[(size_t)q] : nullptr;
        tg.last[t] = m->plog.last ? R.last[(size_t)q] : nullptr;
    }
    {
        const int t = tg.count++;
        tg.pivot[t] = c;
        tg.rate[t] = m->rate; tg.next[t] = m->next; tg.hops[t] = m->hops; tg.last[t] = m->plog.last;
    }
    int64_t *d_index = R.idx;
    FWX_HIP(hipMemcpyAsync(d_index, index, (size_t)count * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(replay_entries_kernel<T>, dim3((unsigned)count), dim3(64), 0, s, d_index, n, R.ld, 0, c,
                       (const T *)m->rate0, m->next ? m->next0 : nullptr, m->hops ? m->hops0 : nullptr,
                       (const T *)R.w, (const T *)R.ct, m->next ? R.cnt : nullptr, R.wh, R.cht, tg);
    FWX_HIP(hipGetLastError());
    return FWX_OK;
}

}  // namespace

extern "C" {

int fwx_abi_version(void) { return FWX_ABI_VERSION; }

int fwx_device_count(void) { return device_count(); }

int fwx_last_hip_error(void) { return g_last_hip; }

int fwx_hip_versions(int32_t *built_against, int32_t *runtime)
{
    if (built_against) *built_against = HIP_VERSION;
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) {
        (void)hipGetLastError();
        v = 0;
    }
    if (runtime) *runtime = v;
    // same major.minor: HIP_VERSION = major * 10^7 + minor * 10^5 + patch
    return (v / 100000 == HIP_VERSION / 100000) ? 1 : 0;
}

int fwx_test_fail_after(int32_t countdown)
{
    g_fail_countdown = countdown > 0 ? countdown : 0;
    return FWX_OK;
}

const char *fwx_strerror(int status)
{
    switch (status) {
    case FWX_OK: return "ok";
    case FWX_ERR_INVALID: return "invalid argument";
    case FWX_ERR_NO_DEVICE: return "no HIP device visible (libfwx has no CPU fallback)";
    case FWX_ERR_HIP: return "HIP runtime error";
    case FWX_ERR_OOM: return "out of memory";
    case FWX_ERR_CYCLE: return "next-hop walk does not reach the destination (cycle)";
    case FWX_ERR_CAPACITY: return "output buffer too small";
    case FWX_ERR_UNSUPPORTED: return "unsupported option combination";
    case FWX_ERR_RCCL: return "RCCL could not be loaded or an RCCL call failed";
    case FWX_ERR_INTERNAL: return "internal error (an exception was stopped at the C ABI)";
    default: return "unknown status";
    }
}

int fwx_solve_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, const fwx_opts *opts)
{
    return fwxi::guarded([&]() -> int { return solve_host<double>(n, rate, next, hops, opts); });
}

int fwx_solve_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, const fwx_opts *opts)
{
    return fwxi::guarded([&]() -> int { return solve_host<float>(n, rate, next, hops, opts); });
}

int fwx_follow_path(int32_t n, const int32_t *next, int32_t src, int32_t dst, int32_t *out,
                    int32_t cap)
{
    return fwxi::guarded([&]() -> int {
        if (n < 0 || !next || src < 0 || dst < 0 || src >= n || dst >= n || cap < 0 ||
            (cap > 0 && !out))
            return FWX_ERR_INVALID;
        const size_t N = (size_t)n;
        if (next[(size_t)src * N + dst] < 0) return 0;
        int32_t len = 0, cur = src;
        while (cur != dst || len == 0) {
            const int32_t nx = next[(size_t)cur * N + dst];
            if (nx < 0 || nx >= n || len >= n) return FWX_ERR_CYCLE;
            if (len >= cap) return FWX_ERR_CAPACITY;
            out[len++] = nx;
            cur = nx;
        }
        return len;
    });
}

int fwx_matrix_create(fwx_matrix **out, int32_t n, int32_t dtype, int32_t with_next,
                      int32_t with_hops, int32_t device)
{
    return fwxi::guarded([&]() -> int {
        if (!out || n < 0 || (dtype != FWX_F32 && dtype != FWX_F64) || (with_hops && !with_next))
            return FWX_ERR_INVALID;
        *out = nullptr;
        DeviceGuard g;
        int rc = g.enter(device);
        if (rc) return rc;
        int dev = 0;
        FWX_HIP(hipGetDevice(&dev));
        fwx_matrix *m = new (std::nothrow) fwx_matrix();
        if (!m) return FWX_ERR_OOM;
        memset(m, 0, sizeof(*m));
        m->n = n; m->dtype = dtype; m->device = dev;
        const size_t es = dtype == FWX_F64 ? 8 : 4;
        // rows of 16-byte vectors for any n (fwx_matrix::nd): the arrays are nd x nd, the padding is
        // written here, once -- nothing ever stores a different value into it
        const int vw = (int)(16 / es);
        m->nd = (n + vw - 1) / vw * vw;
        const size_t nn = (size_t)m->nd * (size_t)m->nd;
        hipError_t e = hipMalloc(&m->rate, nn * es ? nn * es : 1);
        if (e == hipSuccess && with_next) e = hipMalloc((void **)&m->next, nn * 4 ? nn * 4 : 1);
        if (e == hipSuccess && with_hops) e = hipMalloc((void **)&m->hops, nn * 4 ? nn * 4 : 1);
        if (e == hipSuccess) e = hipMalloc((void **)&m->scratch, ((size_t)n + 2) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&m->upd, FWX_UPDATE_SHARDS * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&m->flag, 16);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e == hipSuccess && m->nd != n) {
            e = hipMemsetAsync(m->rate, 0, nn * es, m->stream);                                  // +0.0
            if (e == hipSuccess && m->next) e = hipMemsetAsync(m->next, 0xFF, nn * 4, m->stream);   // -1
            if (e == hipSuccess && m->hops) e = hipMemsetAsync(m->hops, 0, nn * 4, m->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
        }
        if (e != hipSuccess) {
            g_last_hip = (int)e;
            (void)hipGetLastError();
            fwx_matrix_destroy(m);
            return e == hipErrorOutOfMemory ? FWX_ERR_OOM : FWX_ERR_HIP;
        }
        *out = m;
        return FWX_OK;
    });
}

int fwx_matrix_destroy(fwx_matrix *m)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_OK;
        if (m->multi) {
            multi_destroy(m);
            delete m;
            return FWX_OK;
        }
        DeviceGuard g;
        (void)g.enter(m->device);
        // order: retire every command that used the arrays, then the streams and their events, then
        // the memory
        drain_stream(m->stream);
        if (m->side) m->side->drain();
        delete m->side;
        m->side = nullptr;
        if (m->stream) (void)hipStreamDestroy(m->stream);
        m->stream = nullptr;
        if (m->rate) (void)hipFree(m->rate);
        if (m->next) (void)hipFree(m->next);
        if (m->hops) (void)hipFree(m->hops);
        if (m->scratch) (void)hipFree(m->scratch);
        if (m->upd) (void)hipFree(m->upd);
        if (m->plog.last) (void)hipFree(m->plog.last);
        if (m->plog.at_col) (void)hipFree(m->plog.at_col);
        if (m->plog.at_row) (void)hipFree(m->plog.at_row);
        if (m->next0) (void)hipFree(m->next0);
        if (m->rate0) (void)hipFree(m->rate0);
        if (m->hops0) (void)hipFree(m->hops0);
        if (m->walk) (void)hipFree(m->walk);
        resume_free(m->resume);
        m->resume = nullptr;
        if (m->ws) (void)hipFree(m->ws);
        if (m->flag) (void)hipFree(m->flag);
        delete m;
        return FWX_OK;
    });
}

int fwx_matrix_upload(fwx_matrix *m, const void *rate, const int32_t *next, const int32_t *hops)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (m->n == 0) return FWX_OK;
        if (!rate || (m->next && !next) || (m->hops && !hops)) return FWX_ERR_INVALID;
        if (m->multi) return multi_upload(m, rate, next, hops);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t nn = (size_t)m->nd * (size_t)m->nd, es = m->dtype == FWX_F64 ? 8 : 4;
        // hipMemcpyDefault: the sources may be host arrays (what an FFI hands over) or device arrays
        // (a caller that keeps its pristine input in HBM, e.g. the benchmark)
        hipStream_t s = m->stream;
        m->dom_known = 0;                  // a new input: the domain check has to look at it
        if (m->resume) m->resume->valid_upto = 0;   // ... and nothing of the old solve can be resumed
        if ((rc = copy_in(m, m->rate, rate, es, s))) return rc;
        if (m->next && (rc = copy_in(m, m->next, next, 4, s))) return rc;
        if (m->hops && (rc = copy_in(m, m->hops, hops, 4, s))) return rc;
        if (m->plog.last || (m->keep && m->next)) {
            // traced matrix: keep the uploaded next-hops (paths of entries never improved); kept input
            FWX_HIP(hipMemcpyAsync(m->next0, m->next, nn * 4, hipMemcpyDeviceToDevice, s));
            m->rec_ready = 0;   // the trace of an earlier input is stale
        }
        if (m->keep) {
            FWX_HIP(hipMemcpyAsync(m->rate0, m->rate, nn * es, hipMemcpyDeviceToDevice, s));
            if (m->hops) FWX_HIP(hipMemcpyAsync(m->hops0, m->hops, nn * 4, hipMemcpyDeviceToDevice, s));
            m->kept_valid = 1;
        }
        FWX_HIP(hipStreamSynchronize(s));
        m->fresh = 1;
        if (m->resume) m->resume->state_at = m->keep ? 0 : -1;
        return FWX_OK;
    });
}

int fwx_matrix_enable_path_log(fwx_matrix *m)
{
    return fwxi::guarded([&]() -> int {
        if (!m || !m->next || m->plog.last) return FWX_ERR_INVALID;
        if (m->resume) return FWX_ERR_INVALID;     // the checkpoints were sized without the trace: enable it first
        if (m->n == 0) return FWX_OK;
        if (m->multi) return multi_enable_path_log(m);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t nn = (size_t)m->nd * (size_t)m->nd;
        FWX_HIP(hipMalloc((void **)&m->plog.at_col, nn * 4));
        FWX_HIP(hipMalloc((void **)&m->plog.at_row, nn * 4));
        if (!m->next0) FWX_HIP(hipMalloc((void **)&m->next0, nn * 4));
        FWX_HIP(hipMalloc((void **)&m->plog.last, nn * 4));      // last: `last != nullptr` = enabled
        // next0 = the UPLOADED next-hops.  If the arrays already hold an unsolved upload, keep it;
        // otherwise (nothing uploaded yet, or already solved) `fresh` is 0 and a traced solve is
        // refused until the next upload, which fills next0.
        if (m->fresh) {
            FWX_HIP(hipMemcpyAsync(m->next0, m->next, nn * 4, hipMemcpyDeviceToDevice, m->stream));
            FWX_HIP(hipStreamSynchronize(m->stream));
        }
        m->rec_ready = 0;
        return FWX_OK;
    });
}

int fwx_matrix_keep_input(fwx_matrix *m)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (m->keep || m->n == 0) return FWX_OK;
        if (m->multi) return multi_keep_input(m);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t nn = (size_t)m->nd * (size_t)m->nd, es = m->dtype == FWX_F64 ? 8 : 4;
        FWX_HIP(hipMalloc(&m->rate0, nn * es));
        if (m->next && !m->next0) FWX_HIP(hipMalloc((void **)&m->next0, nn * 4));
        if (m->hops) FWX_HIP(hipMalloc((void **)&m->hops0, nn * 4));
        m->keep = 1;
        if (m->fresh) {          // an unsolved upload is in the arrays: that is the input to keep
            hipStream_t s = m->stream;
            FWX_HIP(hipMemcpyAsync(m->rate0, m->rate, nn * es, hipMemcpyDeviceToDevice, s));
            if (m->next) FWX_HIP(hipMemcpyAsync(m->next0, m->next, nn * 4, hipMemcpyDeviceToDevice, s));
            if (m->hops) FWX_HIP(hipMemcpyAsync(m->hops0, m->hops, nn * 4, hipMemcpyDeviceToDevice, s));
            FWX_HIP(hipStreamSynchronize(s));
            m->kept_valid = 1;
        }
        return FWX_OK;
    });
}

int fwx_matrix_patch_input(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                           const int32_t *next_vals, const int32_t *hops_vals)
{
    return fwxi::guarded([&]() -> int {
        if (!m || count < 0 || count > FWX_MAX_PATCH || (count > 0 && (!index || !rate_vals)))
            return FWX_ERR_INVALID;
        if (!m->keep || !m->kept_valid) return FWX_ERR_INVALID;
        if ((next_vals && !m->next) || (hops_vals && !m->hops)) return FWX_ERR_INVALID;
        const int64_t nn64 = (int64_t)m->n * m->n;
        for (int32_t q = 0; q < count; ++q)
            if (index[q] < 0 || index[q] >= nn64) return FWX_ERR_INVALID;
        if (m->multi) return multi_patch_input(m, count, index, rate_vals, next_vals, hops_vals);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t nn = (size_t)m->nd * (size_t)m->nd, es = m->dtype == FWX_F64 ? 8 : 4;
        hipStream_t s = m->stream;
        if (m->resume) m->resume->valid_upto = 0;   // the kept input changes without a replay
        // the remembered domain answer survives a patch whose values are themselves inside the domain
        // (rate >= +0 and not NaN; a non-zero rate comes with a next-hop >= 0); anything else, or a
        // non-zero rate patched in without its next-hop, sends the next solve through the check again
        if (m->dom_known && !patch_keeps_domain(m, count, rate_vals, next_vals)) m->dom_known = 0;
        for (int32_t q = 0; q < count; ++q) {        // a handful of entries: plain small copies
            const size_t off = dev_offset(m, index[q]);
            FWX_HIP(hipMemcpyAsync((char *)m->rate0 + off * es, (const char *)rate_vals + (size_t)q * es, es,
                                   hipMemcpyHostToDevice, s));
            if (next_vals) FWX_HIP(hipMemcpyAsync(m->next0 + off, next_vals + q, 4, hipMemcpyHostToDevice, s));
            if (hops_vals) FWX_HIP(hipMemcpyAsync(m->hops0 + off, hops_vals + q, 4, hipMemcpyHostToDevice, s));
        }
        FWX_HIP(hipMemcpyAsync(m->rate, m->rate0, nn * es, hipMemcpyDeviceToDevice, s));
        if (m->next) FWX_HIP(hipMemcpyAsync(m->next, m->next0, nn * 4, hipMemcpyDeviceToDevice, s));
        if (m->hops) FWX_HIP(hipMemcpyAsync(m->hops, m->hops0, nn * 4, hipMemcpyDeviceToDevice, s));
        FWX_HIP(hipStreamSynchronize(s));
        m->fresh = 1;
        m->rec_ready = 0;
        if (m->resume) m->resume->state_at = 0;      // the patched kept input, unsolved
        return FWX_OK;
    });
}

int fwx_device_memory(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes)
{
    return fwxi::guarded([&]() -> int {
        DeviceGuard g;
        int rc = g.enter(device);
        if (rc) return rc;
        size_t f = 0, t = 0;
        FWX_HIP(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
        return FWX_OK;
    });
}

int fwx_matrix_resume_bytes(const fwx_matrix *m, int32_t checkpoints, uint64_t *bytes_out)
{
    return fwxi::guarded([&]() -> int {
        if (!m || !bytes_out || checkpoints < 0 || checkpoints > FWX_MAX_CHECKPOINTS) return FWX_ERR_INVALID;
        const MultiDims d = resume_dims(m);
        const uint64_t es = m->dtype == FWX_F64 ? 8 : 4;
        // per checkpoint: one copy of every array; panels: w + ct (+ cnt) (+ wh + cht) for all pivots
        // (a partitioned handle: every partition keeps all pivot ROWS, its own part of the columns)
        const uint64_t per_cp = d.cells * (es + (m->next ? 4 : 0) + (m->hops ? 4 : 0) + (m->plog.last ? 12 : 0));
        const uint64_t panels = d.w_cells * es + d.col_cells * es + (m->next ? d.col_cells * 4 : 0) +
                                (m->hops ? d.w_cells * 4 + d.col_cells * 4 : 0);
        *bytes_out = (uint64_t)checkpoints * per_cp + panels + (uint64_t)FWX_MAX_PATCH * 8;
        return FWX_OK;
    });
}

int fwx_matrix_enable_resume(fwx_matrix *m, int32_t checkpoints)
{
    return fwxi::guarded([&]() -> int {
        if (!m || checkpoints < 1 || checkpoints > FWX_MAX_CHECKPOINTS) return FWX_ERR_INVALID;
        if (m->resume) return FWX_ERR_INVALID;
        if (!m->keep) return FWX_ERR_INVALID;                // replays start from the kept input
        if (m->multi) return m->n <= kSmallSolveAutoMax ? FWX_ERR_UNSUPPORTED : multi_enable_resume(m, checkpoints);
        const int n = m->nd;                                 // everything below is sized like the device arrays
        const bool f64 = m->dtype == FWX_F64;
        // resumable = AUTO takes the fused engine for this order (fwx.h fwx_engine)
        if (m->n <= kSmallSolveAutoMax) return FWX_ERR_UNSUPPORTED;
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        fail_point();
        // a partial set (an allocation failed, or threw) is released again: the handle stays usable
        struct Holder {
            Resume *r = new Resume();
            ~Holder() { resume_free(r); }
        } hold;
        Resume *R = hold.r;
        for (auto *v : {&R->next, &R->hops, &R->last, &R->at_col, &R->at_row}) v->reserve((size_t)checkpoints);
        R->rate.reserve((size_t)checkpoints);
        R->pivot.reserve((size_t)checkpoints);
        const size_t es = f64 ? 8 : 4, nn = (size_t)n * n;
        R->ld = (n + 3) & ~3;
        // checkpoints at the multiples of 64 closest to q * n / (checkpoints + 1)
        for (int q = 1; q <= checkpoints; ++q) {
            int p = (int)(((int64_t)m->n * q / (checkpoints + 1) + 32) / 64 * 64);
            if (p <= 0 || p >= m->n || (!R->pivot.empty() && p <= R->pivot.back())) continue;
            R->pivot.push_back(p);
        }
        R->count = (int)R->pivot.size();
        auto alloc = [&](void **p, size_t bytes) -> int { FWX_HIP(hipMalloc(p, bytes)); return FWX_OK; };
        for (int q = 0; q < R->count; ++q) {
            void *p = nullptr;
            if ((rc = alloc(&p, nn * es))) return rc;
            R->rate.push_back(p);
            if (m->next) { if ((rc = alloc(&p, nn * 4))) return rc; R->next.push_back((int32_t *)p); }
            if (m->hops) { if ((rc = alloc(&p, nn * 4))) return rc; R->hops.push_back((int32_t *)p); }
            if (m->plog.last)
                for (auto *v : {&R->last, &R->at_col, &R->at_row}) {
                    if ((rc = alloc(&p, nn * 4))) return rc;
                    v->push_back((int32_t *)p);
                }
        }
        const size_t pan = (size_t)n * R->ld;
        if ((rc = alloc(&R->w, nn * es)) || (rc = alloc(&R->ct, pan * es))) return rc;
        if (m->next && (rc = alloc((void **)&R->cnt, pan * 4))) return rc;
        if (m->hops && ((rc = alloc((void **)&R->wh, nn * 4)) || (rc = alloc((void **)&R->cht, pan * 4)))) return rc;
        if ((rc = alloc((void **)&R->idx, (size_t)FWX_MAX_PATCH * 8))) return rc;
        R->state_at = (m->fresh && m->kept_valid) ? 0 : -1;
        m->resume = R;            // complete: owned by the handle from here on
        hold.r = nullptr;
        return R->count;
    });
}

int fwx_matrix_resolve(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                       const int32_t *next_vals, const int32_t *hops_vals, const fwx_opts *opts,
                       int32_t *resumed_from)
{
    return fwxi::guarded([&]() -> int {
        if (resumed_from) *resumed_from = 0;
        if (!m || count < 0 || count > FWX_MAX_PATCH || (count > 0 && (!index || !rate_vals)))
            return FWX_ERR_INVALID;
        if (!m->keep || !m->kept_valid) return FWX_ERR_INVALID;
        if ((next_vals && !m->next) || (hops_vals && !m->hops)) return FWX_ERR_INVALID;
        const int64_t nn64 = (int64_t)m->n * m->n;
        int64_t lowest = m->n;                       // min over the patched entries' indices
        for (int32_t q = 0; q < count; ++q) {
            if (index[q] < 0 || index[q] >= nn64) return FWX_ERR_INVALID;
            const int64_t i = index[q] / m->n, j = index[q] % m->n;
            lowest = i < lowest ? i : lowest;
            lowest = j < lowest ? j : lowest;
        }
        Opts op;
        int rc = read_opts(opts, m->n, op);
        if (rc) return rc;
        // the checkpoint to resume from: the last one at or before `lowest` that still belongs to the
        // solve of the kept input -- on a handle that records them, on the reference's domain
        // (the fused engine ran and will run again), for a whole-range uncounted solve
        int c_idx = -1;
        Resume *R = m->resume;
        if (R && count > 0 && m->dom_known && !op.updates_out && !op.has_stream && op.k_begin == 0 &&
            op.k_end == m->n && op.engine != FWX_ENGINE_PERK)
            for (int q = 0; q < R->count; ++q)
                if (R->pivot[(size_t)q] <= lowest && R->pivot[(size_t)q] <= R->valid_upto) c_idx = q;
        if (c_idx >= 0) {
            // are the patched values inside the domain?  (fwx_matrix_patch_input keeps dom_known only then)
            if (!patch_keeps_domain(m, count, rate_vals, next_vals)) c_idx = -1;
        }
        if (c_idx < 0) {
            // nothing to resume from: the patched input from pivot 0 (which records anew)
            if ((rc = fwx_matrix_patch_input(m, count, index, rate_vals, next_vals, hops_vals))) return rc;
            return fwx_matrix_solve(m, opts);
        }
        if (m->multi) {
            rc = multi_resolve(m, count, index, rate_vals, next_vals, hops_vals, c_idx, op);
            if (!rc && resumed_from) *resumed_from = R->pivot[(size_t)c_idx];
            return rc;
        }
        DeviceGuard g;
        if ((rc = g.enter(m->device))) return rc;
        hipStream_t s = m->stream;
        const size_t es = m->dtype == FWX_F64 ? 8 : 4;
        const int c = R->pivot[(size_t)c_idx];
        R->valid_upto = 0;                           // until the resumed solve has finished
        // Any error return below leaves the handle in a state the next call can start from: nothing of this
        // call still queued on the stream (the small copies read the caller's arrays), nothing resumable,
        // the live arrays no known state of the kept input (the next resolve / patch_input restores them).
        struct Unwind {
            fwx_matrix *m; hipStream_t s; bool armed = true;
            ~Unwind()
            {
                if (!armed) return;
                (void)hipStreamSynchronize(s);
                m->resume->valid_upto = 0;
                m->resume->state_at = -1;
                m->fresh = 0;
                m->rec_ready = 0;
            }
        } unwind{m, s};
        std::vector<int64_t> dindex((size_t)count);  // offsets in the device arrays (pitch nd)
        for (int32_t q = 0; q < count; ++q) dindex[(size_t)q] = (int64_t)dev_offset(m, index[q]);
        for (int32_t q = 0; q < count; ++q) {        // the kept input first: the replay reads it
            const size_t off = (size_t)dindex[(size_t)q];
            FWX_HIP(hipMemcpyAsync((char *)m->rate0 + off * es, (const char *)rate_vals + (size_t)q * es, es,
                                   hipMemcpyHostToDevice, s));
            if (next_vals) FWX_HIP(hipMemcpyAsync(m->next0 + off, next_vals + q, 4, hipMemcpyHostToDevice, s));
            if (hops_vals) FWX_HIP(hipMemcpyAsync(m->hops0 + off, hops_vals + q, 4, hipMemcpyHostToDevice, s));
        }
        rc = m->dtype == FWX_F64 ? resolve_typed<double>(m, count, dindex.data(), c_idx, s)
                                 : resolve_typed<float>(m, count, dindex.data(), c_idx, s);
        if (rc) return rc;
        m->fresh = 0;
        m->rec_ready = 0;
        R->valid_upto = c;                           // panels < c and checkpoints <= c hold for the NEW input;
                                                     // later ones are the old solve's until this one passes them
        op.k_begin = c;
        R->state_at = op.k_begin;                    // the live arrays: the NEW kept input at time c
        if (m->plog.last) {
            rc = logged_solve(m, op, s, true);
        } else {
            rc = m->dtype == FWX_F64 ? matrix_solve_typed<double>(m, op, nullptr, s)
                                     : matrix_solve_typed<float>(m, op, nullptr, s);
            if (!rc) FWX_HIP(hipStreamSynchronize(s));
        }
        if (rc) return rc;
        unwind.armed = false;
        if (resumed_from) *resumed_from = op.k_begin;
        return FWX_OK;
    });
}

int fwx_matrix_path_log_count(fwx_matrix *m, uint64_t *count_out)
{
    return fwxi::guarded([&]() -> int {
        if (!m || !count_out) return FWX_ERR_INVALID;
        *count_out = (m->plog.last && m->rec_ready) ? m->last_u : 0;
        return FWX_OK;
    });
}

int fwx_matrix_query_exact(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out,
                           int32_t *path_out, int32_t cap)
{
    return fwxi::guarded([&]() -> int {
        if (!m || src < 0 || dst < 0 || src >= m->n || dst >= m->n || cap <= 0 || !path_out)
            return FWX_ERR_INVALID;
        if (!m->plog.last) return FWX_ERR_INVALID;
        if (!m->rec_ready) return FWX_ERR_INVALID;             // no traced solve of this upload yet
        if (m->multi) return multi_query_exact(m, src, dst, rate_out, path_out, cap);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t idx = (size_t)src * m->nd + dst;
        hipStream_t s = m->stream;
        float f32_rate = 0;
        if (rate_out) {
            if (m->dtype == FWX_F64)
                FWX_HIP(hipMemcpyAsync(rate_out, (double *)m->rate + idx, 8, hipMemcpyDeviceToHost, s));
            else
                FWX_HIP(hipMemcpyAsync(&f32_rate, (float *)m->rate + idx, 4, hipMemcpyDeviceToHost, s));
        }
        if (!m->walk || m->walk_cap < cap) {      // grow-only scratch, reused across queries
            if (m->walk) { drain_stream(s); (void)hipFree(m->walk); m->walk = nullptr; }
            FWX_HIP(hipMalloc((void **)&m->walk, ((size_t)4 * cap + 1) * 4));
            m->walk_cap = cap;
        }
        int32_t *len_dev = m->walk + (size_t)4 * cap;
        hipLaunchKernelGGL(exact_path_kernel, dim3(1), dim3(1), 0, s, m->plog, m->next0, m->nd, src,
                           dst, m->walk, cap, len_dev);
        FWX_HIP(hipGetLastError());
        int32_t len = 0;
        FWX_HIP(hipMemcpyAsync(&len, len_dev, 4, hipMemcpyDeviceToHost, s));
        FWX_HIP(hipStreamSynchronize(s));
        if (rate_out && m->dtype != FWX_F64) *rate_out = (double)f32_rate;
        if (len > 0) {
            FWX_HIP(hipMemcpyAsync(path_out, m->walk, (size_t)len * 4, hipMemcpyDeviceToHost, s));
            FWX_HIP(hipStreamSynchronize(s));
        }
        return len;
    });
}

int fwx_matrix_query_exact_batch(fwx_matrix *m, int32_t count, const int32_t *src, const int32_t *dst,
                                 int32_t *len_out, int32_t *path_out, int32_t cap)
{
    return fwxi::guarded([&]() -> int {
        if (!m || count < 0 || cap <= 0) return FWX_ERR_INVALID;
        if (count == 0) return FWX_OK;
        if (!src || !dst || !len_out || !path_out || !m->plog.last) return FWX_ERR_INVALID;
        if (!m->rec_ready) return FWX_ERR_INVALID;
        if (m->multi) return multi_query_exact_batch(m, count, src, dst, len_out, path_out, cap);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        // device scratch from a pooled per-call context: no hipMalloc / hipFree per query (and no hipFree
        // right behind the kernel that used the memory: drain_stream in fwx_internal.h)
        CtxLease lease;
        if ((rc = lease.open())) return rc;
        lease.c->uses_stream(m->stream);     // the kernel below runs on the handle's stream
        const size_t c = (size_t)count;
        struct { void *p = nullptr; } d_src, d_dst, d_len, d_paths, d_stacks;
        void *ids = nullptr;
        if ((rc = lease.c->reserve(CallCtx::NEXT, c * 12, &ids)) ||
            (rc = lease.c->reserve(CallCtx::RATE, c * cap * 4, &d_paths.p)) ||
            (rc = lease.c->reserve(CallCtx::WS, c * cap * 12, &d_stacks.p)))
            return rc;
        d_src.p = ids;
        d_dst.p = (char *)ids + c * 4;
        d_len.p = (char *)ids + c * 8;
        hipStream_t s = m->stream;
        FWX_HIP(hipMemcpyAsync(d_src.p, src, c * 4, hipMemcpyHostToDevice, s));
        FWX_HIP(hipMemcpyAsync(d_dst.p, dst, c * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(exact_paths_kernel, dim3((unsigned)((c + 63) / 64)), dim3(64), 0, s, m->plog,
                           m->next0, m->n, m->nd, count, (const int32_t *)d_src.p, (const int32_t *)d_dst.p,
                           (int32_t *)d_paths.p, (int32_t *)d_stacks.p, cap, (int32_t *)d_len.p);
        FWX_HIP(hipGetLastError());
        FWX_HIP(hipMemcpyAsync(len_out, d_len.p, c * 4, hipMemcpyDeviceToHost, s));
        FWX_HIP(hipMemcpyAsync(path_out, d_paths.p, c * cap * 4, hipMemcpyDeviceToHost, s));
        FWX_HIP(hipStreamSynchronize(s));
        return FWX_OK;
    });
}

int fwx_matrix_download(fwx_matrix *m, void *rate, int32_t *next, int32_t *hops)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (m->n == 0) return FWX_OK;
        if ((next && !m->next) || (hops && !m->hops)) return FWX_ERR_INVALID;
        if (m->multi) return multi_download(m, rate, next, hops);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t es = m->dtype == FWX_F64 ? 8 : 4;
        hipStream_t s = m->stream;
        if (rate && (rc = copy_out(m, rate, m->rate, es, s))) return rc;
        if (next && (rc = copy_out(m, next, m->next, 4, s))) return rc;
        if (hops && (rc = copy_out(m, hops, m->hops, 4, s))) return rc;
        FWX_HIP(hipStreamSynchronize(s));
        return FWX_OK;
    });
}

int fwx_matrix_solve(fwx_matrix *m, const fwx_opts *opts)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (m->n == 0) return FWX_OK;
        Opts op;
        int rc = read_opts(opts, m->n, op);
        if (rc) return rc;
        if (m->multi) return multi_solve(m, op);
        DeviceGuard g;
        if ((rc = g.enter(m->device))) return rc;
        hipStream_t s = op.has_stream ? op.stream : m->stream;
        if (m->plog.last) return logged_solve(m, op, s);
        unsigned long long *upd = op.updates_out ? m->upd : nullptr;
        if (upd) FWX_HIP(hipMemsetAsync(upd, 0, FWX_UPDATE_SHARDS * 8, s));
        if (m->dtype == FWX_F64)
            rc = matrix_solve_typed<double>(m, op, upd, s);
        else
            rc = matrix_solve_typed<float>(m, op, upd, s);
        if (rc) return rc;
        FWX_HIP(hipStreamSynchronize(s));
        if (op.has_stream) drain_stream(s);   // the caller may destroy the handle next: nothing of this
                                              // solve is left un-retired on a stream the handle does not own
        if (upd) return sum_updates(upd, op.updates_out, s);
        return FWX_OK;
    });
}

int fwx_matrix_query(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out, int32_t *path_out,
                     int32_t cap)
{
    return fwxi::guarded([&]() -> int {
        if (!m || src < 0 || dst < 0 || src >= m->n || dst >= m->n || cap < 0 || (cap > 0 && !path_out))
            return FWX_ERR_INVALID;
        if (m->multi) return multi_query(m, src, dst, rate_out, path_out, cap);
        DeviceGuard g;
        int rc = g.enter(m->device);
        if (rc) return rc;
        const size_t idx = (size_t)src * m->nd + dst;
        hipStream_t s = m->stream;
        float f32_rate = 0;
        if (rate_out) {
            if (m->dtype == FWX_F64)
                FWX_HIP(hipMemcpyAsync(rate_out, (double *)m->rate + idx, 8, hipMemcpyDeviceToHost, s));
            else
                FWX_HIP(hipMemcpyAsync(&f32_rate, (float *)m->rate + idx, 4, hipMemcpyDeviceToHost, s));
        }
        if (!m->next) {
            FWX_HIP(hipStreamSynchronize(s));
            if (rate_out && m->dtype != FWX_F64) *rate_out = (double)f32_rate;
            return FWX_ERR_INVALID;
        }
        const int dcap = cap < m->n ? cap : m->n;
        hipLaunchKernelGGL(follow_path_kernel, dim3(1), dim3(1), 0, s, m->next, m->n, m->nd, src, dst,
                           m->scratch + 1, dcap, m->scratch);
        FWX_HIP(hipGetLastError());
        int32_t len = 0;
        FWX_HIP(hipMemcpyAsync(&len, m->scratch, 4, hipMemcpyDeviceToHost, s));
        FWX_HIP(hipStreamSynchronize(s));
        if (rate_out && m->dtype != FWX_F64) *rate_out = (double)f32_rate;
        if (len > 0) {
            FWX_HIP(hipMemcpyAsync(path_out, m->scratch + 1, (size_t)len * 4, hipMemcpyDeviceToHost, s));
            FWX_HIP(hipStreamSynchronize(s));
        }
        return len;
    });
}

int fwx_dev_relax(const fwx_slab *slab, const fwx_pivots *piv, int32_t serpentine,
                  unsigned long long *d_updates, void *stream)
{
    return fwxi::guarded([&]() -> int {
        return fwx_dev_relax_skip(slab, piv, serpentine, d_updates, 0, 0, stream);
    });
}

int fwx_dev_relax_skip(const fwx_slab *slab, const fwx_pivots *piv, int32_t serpentine,
                       unsigned long long *d_updates, int32_t skip_lo, int32_t skip_hi, void *stream)
{
    return fwxi::guarded([&]() -> int {
        int rc = check_slab(slab);
        if (rc) return rc;
        if (!piv || piv->k_begin < 0 || piv->k_end < piv->k_begin || piv->k_end > slab->n)
            return FWX_ERR_INVALID;
        if (slab->rows == 0 || slab->n == 0 || piv->k_end == piv->k_begin) return FWX_OK;
        if (!piv->rate || (slab->hops && !piv->hops)) return FWX_ERR_INVALID;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        hipStream_t s = (hipStream_t)stream;
        if (skip_lo < 0 || skip_hi < skip_lo || skip_hi > slab->rows ||
            (skip_hi > skip_lo && (skip_lo % 4 || skip_hi % 4)))
            return FWX_ERR_INVALID;
        if (slab->dtype == FWX_F64)
            return relax_range<double>((double *)slab->rate, slab->next, slab->hops, slab->rows,
                                       slab->n, slab->row0, (const double *)piv->rate, piv->hops,
                                       piv->stride, piv->k_begin, piv->k_end, serpentine, d_updates, s,
                                       fwx::PathLog(), skip_lo, skip_hi, slab->next ? piv->next : nullptr);
        return relax_range<float>((float *)slab->rate, slab->next, slab->hops, slab->rows, slab->n,
                                  slab->row0, (const float *)piv->rate, piv->hops, piv->stride,
                                  piv->k_begin, piv->k_end, serpentine, d_updates, s, fwx::PathLog(),
                                  skip_lo, skip_hi, slab->next ? piv->next : nullptr);
    });
}

int fwx_dev_panel(const fwx_slab *block, void *w_rate, int32_t *w_hops,
                  unsigned long long *d_updates, void *stream)
{
    return fwxi::guarded([&]() -> int {
        int rc = check_slab(block);
        if (rc) return rc;
        if (block->rows == 0 || block->n == 0) return FWX_OK;
        if (!w_rate || (block->hops && !w_hops)) return FWX_ERR_INVALID;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        hipStream_t s = (hipStream_t)stream;
        if (block->dtype == FWX_F64)
            return panel_impl<double>(block, (double *)w_rate, w_hops, d_updates, s);
        return panel_impl<float>(block, (float *)w_rate, w_hops, d_updates, s);
    });
}

int fwx_dev_solve(const fwx_slab *full, const fwx_opts *opts)
{
    return fwxi::guarded([&]() -> int {
        int rc = check_slab(full);
        if (rc) return rc;
        if (full->row0 != 0 || full->rows != full->n) return FWX_ERR_INVALID;
        if (full->n == 0) return FWX_OK;
        Opts op;
        if ((rc = read_opts(opts, full->n, op))) return rc;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        fwx_matrix m;
        memset(&m, 0, sizeof(m));
        m.n = m.nd = full->n; m.dtype = full->dtype;      // caller-owned memory: pitch n, no padding
        m.rate = full->rate; m.next = full->next; m.hops = full->hops;
        DeviceGuard g;                       // the context belongs to the device the call runs on
        if ((rc = g.enter(op.device))) return rc;
        CtxLease lease;
        if ((rc = lease.open())) return rc;
        hipStream_t s = op.has_stream ? op.stream : lease.c->s;
        if (op.has_stream) lease.c->uses_stream(op.stream);
        struct { void *p = nullptr; } upd;
        if (op.updates_out) {
            void *small = nullptr;
            if ((rc = lease.c->reserve(CallCtx::SMALL, (FWX_UPDATE_SHARDS + 2) * sizeof(unsigned long long), &small))) return rc;
            upd.p = small;
            FWX_HIP(hipMemsetAsync(upd.p, 0, FWX_UPDATE_SHARDS * 8, s));
        }
        rc = full->dtype == FWX_F64
                 ? matrix_solve_typed<double>(&m, op, (unsigned long long *)upd.p, s, lease.c)
                 : matrix_solve_typed<float>(&m, op, (unsigned long long *)upd.p, s, lease.c);
        if (rc) return rc;
        FWX_HIP(hipStreamSynchronize(s));
        if (op.updates_out) return sum_updates((unsigned long long *)upd.p, op.updates_out, s);
        return FWX_OK;
    });
}

int fwx_dev_follow_paths(int32_t n, const int32_t *next, int32_t count, const int32_t *src,
                         const int32_t *dst, int32_t *len_out, const void *edge_rate,
                         int32_t dtype, double *prod_out, int32_t *path_out, int32_t cap,
                         void *stream)
{
    return fwxi::guarded([&]() -> int {
        if (n < 0 || count < 0 || cap < 0 || (dtype != FWX_F32 && dtype != FWX_F64))
            return FWX_ERR_INVALID;
        if (count == 0) return FWX_OK;
        if (!next || !src || !dst || !len_out || (prod_out && !edge_rate) || (path_out && cap == 0))
            return FWX_ERR_INVALID;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        hipStream_t s = (hipStream_t)stream;
        const dim3 grid((unsigned)((count + 255) / 256)), block(256);
        if (dtype == FWX_F64)
            hipLaunchKernelGGL(follow_paths_kernel<double>, grid, block, 0, s, next, n, count, src,
                               dst, len_out, (const double *)edge_rate, prod_out, path_out, cap);
        else
            hipLaunchKernelGGL(follow_paths_kernel<float>, grid, block, 0, s, next, n, count, src, dst,
                               len_out, (const float *)edge_rate, prod_out, path_out, cap);
        FWX_HIP(hipGetLastError());
        return FWX_OK;
    });
}

int fwx_dev_panel_snap(const fwx_slab *block, void *w_rate, int32_t *w_hops, const fwx_trace *trace,
                       void *stream)
{
    return fwxi::guarded([&]() -> int {
        int rc = check_slab(block);
        if (rc) return rc;
        if (block->rows == 0 || block->n == 0) return FWX_OK;
        if (!w_rate || block->rows > FWX_FUSED_B || (block->hops && !w_hops)) return FWX_ERR_INVALID;
        if (trace && (!trace->last || !trace->at_row || !block->next)) return FWX_ERR_INVALID;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        hipStream_t s = (hipStream_t)stream;
        fwx::PathLog pl = fwx::PathLog();
        if (trace) { pl.last = trace->last; pl.at_col = trace->at_col; pl.at_row = trace->at_row; }
        if (block->dtype == FWX_F64)
            FWX_HIP(fwx::launch_fused_panel<double>((const double *)block->rate, block->n, block->row0,
                                                    block->rows, (double *)w_rate, s, pl, block->hops, w_hops));
        else
            FWX_HIP(fwx::launch_fused_panel<float>((const float *)block->rate, block->n, block->row0,
                                                   block->rows, (float *)w_rate, s, pl, block->hops, w_hops));
        return FWX_OK;
    });
}

int fwx_dev_check_nonneg(const fwx_slab *slab, int32_t *d_flag, void *stream)
{
    return fwxi::guarded([&]() -> int {
        int rc = check_slab(slab);
        if (rc) return rc;
        if (!d_flag) return FWX_ERR_INVALID;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        hipStream_t s = (hipStream_t)stream;
        if (slab->dtype == FWX_F64)
            FWX_HIP(fwx::launch_nonneg_check((const double *)slab->rate, slab->next,
                                             (size_t)slab->rows * slab->n, (int *)d_flag, s));
        else
            FWX_HIP(fwx::launch_nonneg_check((const float *)slab->rate, slab->next,
                                             (size_t)slab->rows * slab->n, (int *)d_flag, s));
        return FWX_OK;
    });
}

int fwx_dev_relax_fused(const fwx_slab *slab, const fwx_pivots *piv, const fwx_fused_scratch *scratch,
                        const fwx_trace *trace, unsigned long long *d_updates, int32_t flags,
                        void *stream)
{
    return fwxi::guarded([&]() -> int {
        return fwx_dev_relax_fused_skip(slab, piv, scratch, trace, d_updates, flags, 0, 0, stream);
    });
}

int fwx_dev_relax_fused_skip(const fwx_slab *slab, const fwx_pivots *piv,
                             const fwx_fused_scratch *scratch, const fwx_trace *trace,
                             unsigned long long *d_updates, int32_t flags, int32_t skip_lo,
                             int32_t skip_hi, void *stream)
{
    return fwxi::guarded([&]() -> int {
        int rc = check_slab(slab);
        if (rc) return rc;
        if (!piv || piv->k_begin < 0 || piv->k_end < piv->k_begin || piv->k_end > slab->n ||
            piv->k_end - piv->k_begin > FWX_FUSED_B)
            return FWX_ERR_INVALID;
        if (slab->rows == 0 || slab->n == 0 || piv->k_end == piv->k_begin) return FWX_OK;
        if (!piv->rate || piv->stride != slab->n || !scratch || !scratch->col_rate ||
            (slab->next && !scratch->col_next) || (slab->hops && (!scratch->col_hops || !piv->hops)))
            return FWX_ERR_INVALID;
        if (trace && (!slab->next || !trace->last || !trace->at_col)) return FWX_ERR_INVALID;
        if (skip_lo < 0 || skip_hi < skip_lo || skip_hi > slab->rows ||
            (skip_hi > skip_lo && (skip_lo % 8 || skip_hi % 8)))
            return FWX_ERR_INVALID;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        hipStream_t s = (hipStream_t)stream;
        const bool nonneg = (flags & FWX_FLAG_NONNEG) != 0;
        if (slab->dtype == FWX_F64)
            return fused_block<double>(slab, piv->k_begin, piv->k_end - piv->k_begin, (const double *)piv->rate,
                                       piv->hops, scratch, trace, d_updates, nonneg, s, skip_lo, skip_hi);
        return fused_block<float>(slab, piv->k_begin, piv->k_end - piv->k_begin, (const float *)piv->rate,
                                  piv->hops, scratch, trace, d_updates, nonneg, s, skip_lo, skip_hi);
    });
}

}  // extern "C"
