// fwx_fused.hip -- the fused engine: B = 64 pivots of runAlgo
// (/root/reference/src/lib/Algorithms.hs:42-61) per pass over the matrix, bit-identical to the
// per-k loop.
//
// Why it is exact.  Step k only ever combines an entry with column k and row k AS THEY STAND AT
// THE START OF STEP k (Algorithms.hs:58-60), and neither is modified by step k (:50, :54).  So for
// pivots k0..k0+B-1 an entry's final value is the in-order fold
//       x <- (x < C_k[i] * W_k[j]) ? C_k[i] * W_k[j] : x        k = k0, k0+1, ...
// where W_k = row k at time k and C_k = column k at time k ("time-k snapshots").  Same operands,
// same order, same single multiply and strict compare as the reference: no product is
// re-associated (classic 3-phase blocked Floyd-Warshall is NOT exact: SURVEY.md Appendix B).
// The snapshots themselves are produced by the same fold restricted to the pivot rows/columns:
//
//   fused_rowpanel      every column j of the B pivot rows: exports W[t][j].  Each workgroup evolves the
//                       B x B diagonal block itself, together with its strip
//   fused_colpanel      the B pivot columns of every row i: exports Ct[t][i] (NaN where i == k: skip
//                       i==k) and CNt[t][i] = next_t[i][k0+t].  Reads the block columns of a finished
//                       W (partitioned solves), or -- OWN_D -- evolves the diagonal block itself
//   fused_panels        both of them as ONE launch (single-device solves); fused_panels_next_f32: the
//                       same for f32 + next-hops in 48 VGPRs, so that its 1024-thread workgroups fit
//                       beside two fused_main_arg workgroups on a CU
//   fused_main_max      rates only, f32, inside the domain: the fold as max (two pivots per v_max3_f32),
//                       128 x 128 tile, 8 x 8 entries per thread, one or two passes (64 / 128 pivots) per
//                       launch; fused_main_max_f64 likewise for f64
//   fused_main_arg      rates + next-hops (+ path trace, + hops), f32 inside the domain: max-form fold,
//                       then an arg re-scan of the entries that moved; NP = 1 or 2 complete passes per
//                       launch on the tile kept in registers; fused_main_arg_f64 for f64
//   fused_main          compare form (product, compare, selects): update counting and inputs outside
//                       the domain
//
// Skip set: i==k via NaN in Ct, j==k via NaN injected into the staged W tile, j==i by restoring
// the diagonal entry at write-back.  Scratch copies of diagonal entries may go stale inside the
// panel kernels; they are provably never consumed (every consumer is an i==k or j==k case).
//
// Roofline: per pass 4 B read + 4 B written per entry against 1.5*B = 96 VALU lane-ops per entry in
// the max form (3*B in the compare form): these kernels are VALU-bound, not HBM-bound.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "fwx_kernels.h"

#pragma clang fp contract(off)

namespace fwx {

namespace {

template <typename T> __device__ __forceinline__ T qnan();
template <> __device__ __forceinline__ float qnan<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double qnan<double>() { return __builtin_nan(""); }

// ONE v_max_f32 (the builtin adds a quieting `v_max_f32 x, x, x` in front, as for f64 below; the
// three-operand folds of the main kernels call __builtin_fmaxf twice and get one v_max3_f32).
__device__ __forceinline__ float fmax_t(float a, float b)
{
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// ONE v_max_f64.  __builtin_fmax makes the compiler quiet a possible signalling NaN first -- a second
// `v_max_f64 x, x, x` per relaxation, i.e. three 4.35-cycle f64 instructions where two suffice (the
// f64 fused solve at N = 16384: 513 ms with the builtin).  The instruction itself already is IEEE
// maxNum: max(x, qNaN) = x, and the only NaNs here are the quiet products inf * 0.
__device__ __forceinline__ double fmax_t(double a, double b)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <typename T> struct Vec16;
template <> struct Vec16<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int W = 4;
};
template <> struct Vec16<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int W = 2;
};
template <int W> struct IVec;
template <> struct IVec<4> { typedef int type __attribute__((ext_vector_type(4))); };
template <> struct IVec<2> { typedef int type __attribute__((ext_vector_type(2))); };

constexpr int B = FWX_FUSED_B;

// s_waitcnt immediate (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8):
// all vector-memory loads have returned, the other counters untouched.
#define FWX_WAIT_VMCNT0 0x0F70

// -DFWX_CLOCK_PROBE (a variant build, never shipped): the shader clock the chip holds while the main kernels run --
// every workgroup adds its s_memtime cycles and its s_memrealtime ticks (100 MHz) to two counters that
// fwx_debug_clock reads (tools/measure_clock.py).
#ifdef FWX_CLOCK_PROBE
__device__ unsigned long long g_clk[3];
struct ClockProbe {
    unsigned long long t0, r0;
    __device__ ClockProbe() : t0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()) {}
    __device__ ~ClockProbe()
    {
        if (threadIdx.x == 0) {
            atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - t0);
            atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0);
            atomicAdd(&g_clk[2], 1ull);
        }
    }
};
#define FWX_PROBE ClockProbe probe_
#else
#define FWX_PROBE do { } while (0)
#endif

// Column controls of a main-kernel launch (symmetric look-ahead, fused_range in fwx_api.hip):
// the grid's column tiles start at tile jt0 (a launch over a window of columns), and columns
// [cskip_lo, cskip_hi) (multiples of 4) are left alone -- another launch of the same pass owns them.
struct ColWin {
    int jt0, cskip_lo, cskip_hi;
    int prio = 0;          // FusedArgs::side: the launch's waves issue ahead of the main launch's on a shared SIMD
    __host__ __device__ bool skips(int j) const { return j >= cskip_lo && j < cskip_hi; }
    __host__ __device__ bool clear_of(int j_lo, int j_hi) const { return j_hi <= cskip_lo || j_lo >= cskip_hi; }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), i.e. waits for the snapshot stores each panel step issues to reach memory
// (~1 us per step, 64 steps per panel); nothing in these kernels reads those stores back.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Synchronisation of the panel chain (round 4).  FWX_PANEL_FLAGS=1 (default): no workgroup barrier in the loop --
// the wave on its serial phase raises ONE LDS FLAG PER PIVOT right after publishing that pivot's line, and a later
// wave applies a pivot as soon as its flag is up: the wave that goes serial next has applied all but the last pivot
// of the sub-block before its predecessor is done, so the chain per sub-block is one serial phase + one pivot's
// apply instead of serial phase + barrier + four pivots' apply.  Older sub-blocks are waited for with one flag
// (their last pivot's: lines and flags of a wave go through the LDS queue in order).  =0: the barrier form.
#ifndef FWX_PANEL_FLAGS
#define FWX_PANEL_FLAGS 1
#endif
constexpr int PANEL_FLAG_BYTES = B * 4;
__device__ __forceinline__ void panel_flags_init(int *flag)
{
    if (threadIdx.x < B) flag[threadIdx.x] = 0;
}
__device__ __forceinline__ void panel_flag_raise(int *flag, int t)
{
    // A wave's LDS operations are executed in issue order (one queue per CU; that is what lgkmcnt counts on), so a
    // plain store issued AFTER the lines' stores is seen after them: no wait for their completion on the chain --
    // a release store puts s_waitcnt lgkmcnt(0) in front, ~4 x 64 cycles per sub-block.  The compiler must not
    // move it up: signal fences on both sides.
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __hip_atomic_store(&flag[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
__device__ __forceinline__ void panel_flag_wait(int *flag, int t)
{
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&flag[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0)
        __builtin_amdgcn_s_sleep(1);
}

// ------------------------------------------------------------------------------------------------
// Panel kernels.  One wave per SUB-BLOCK of SB = 4 pivots: 16 waves = 1024 threads per workgroup.
// Wave b owns SB of the 64 panel lines (rows for the row panel, block columns for the column
// panel), one register per line, lanes across the other dimension.  For sub-block b:
//   serial phase  -- wave b alone steps through its SB pivots: the pivot line of each step is its
//                    OWN register (no LDS, no barrier in the chain); it publishes every time-t
//                    line to LDS on the way;
//   apply phase   -- after ONE barrier the later waves fold those SB published pivots into their
//                    own SB lines, in order.
// The serial chain is 64*SB line-updates long, so short sub-blocks (and more waves) shorten it;
// measured at N = 1024 f32: SB = 16 / 8 / 4 -> 2.70 / 2.16 / 2.02 ms per solve.  B/SB barriers per
// panel instead of 64, and no memory latency on the serial chain.  All register
// indices are compile-time constants (loops fully unrolled).
// ------------------------------------------------------------------------------------------------
constexpr int SB = 4;             // pivots per sub-block = lines per wave
constexpr int PANEL_THREADS = (B / SB) * 64;   // one wave per sub-block

template <typename T> __device__ __forceinline__ T readlane(T v, int lane);
template <> __device__ __forceinline__ float readlane<float>(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
template <> __device__ __forceinline__ double readlane<double>(double v, int lane)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <> __device__ __forceinline__ int readlane<int>(int v, int lane)
{
    return __builtin_amdgcn_readlane(v, lane);
}

// Row panel: snapshot panel W of the pivot rows.  Every workgroup evolves the 64 x 64 diagonal
// block itself (lanes = block columns: redundant across workgroups, but it removes a kernel boundary
// and a single-workgroup launch from the critical path) TOGETHER with its own strip of 64 columns,
// in ONE loop over the 16 sub-blocks: a wave holds the diagonal-block part AND the strip part of
// its SB rows, and the pivot-column operand of a row at time t -- D_t[k0+r][k0+t] -- is simply lane
// t of that row's own diagonal-block register (v_readlane), used for both parts.  (Round 1 ran the
// diagonal block and the strip as two loops, 32 barriers, with the column operands going through
// LDS: 35 us per panel at N = 16384 f32; merged: 16 barriers and no LDS round trip for them.)
// HAS_LAST: also keeps the path trace of the pivot rows (PathLog): `last` of each strip entry is
// carried through the 64 pivots beside its rate, and at_row[k0+t][j] = last of (k0+t, j) at time
// k0+t is exported with the snapshot.  last_rows / at_rows point at row k0 of the trace matrices.
// HAS_HOPS: also carries `hops` (= length _path) of the pivot rows: hops' = hops[i][k] + hops[k][j]
// on every successful relaxation (Algorithms.hs:55), and exports wh_out[t][j] = hops of (k0+t, j) at
// time k0+t beside the rate snapshot.  hops_rows points at row k0 of the hops matrix.
// (LDS is carved from one buffer so that the row and the column panel can share a launch:
// fused_panels below)
template <typename T, bool HAS_HOPS> constexpr int rowpanel_lds()
{
    return 2 * B * 64 * (int)sizeof(T) + (HAS_HOPS ? 2 * B * 64 * 4 : 0) + PANEL_FLAG_BYTES;
}
// MAXF (rates only, domain verified by the caller): x <- max(x, c * w) instead of compare + select --
// the same bits on that domain (see fused_main_max; the NaN operands that encode the skip set are
// ignored by max exactly as `x < NaN` is false), one instruction less on the serial chain.
template <typename T, bool HAS_LAST, bool HAS_HOPS, bool MAXF = false, bool FLAGS = FWX_PANEL_FLAGS != 0>
__device__ __forceinline__ void rowpanel_body(char *smem, int bid, const T *rows, int n, int k0, int bt,
                                              T *w_out, const int32_t *last_rows, int32_t *at_rows,
                                              const int32_t *hops_rows, int32_t *wh_out)
{
    typedef T line_t[64];
    typedef int32_t iline_t[64];
    __builtin_amdgcn_s_setprio(3);                 // the panels ARE the serial chain: first in line on their SIMD
    line_t *s_dline = (line_t *)smem;              // published pivot rows, diagonal-block part (time t)
    line_t *s_sline = s_dline + B;                 // ... strip part
    iline_t *s_dh = (iline_t *)(s_sline + B);      // their hops (HAS_HOPS)
    iline_t *s_sh = s_dh + B;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = bid * 64 + lane;
    const bool valid = j < n;
    const int jc = valid ? j : n - 1;

    T pd[SB], ps[SB];                              // this wave's SB rows: diagonal-block / strip columns
    int32_t hd[SB], hs[SB], lp[SB];                // hops of both parts, `last` of the strip part
#pragma unroll
    for (int q = 0; q < SB; ++q) {
        const int r = wave * SB + q;
        pd[q] = (r < bt && lane < bt) ? rows[(size_t)r * n + k0 + lane] : qnan<T>();
        hd[q] = (HAS_HOPS && r < bt && lane < bt) ? hops_rows[(size_t)r * n + k0 + lane] : 0;
        ps[q] = r < bt ? rows[(size_t)r * n + jc] : qnan<T>();
        lp[q] = (HAS_LAST && r < bt) ? last_rows[(size_t)r * n + jc] : -1;
        hs[q] = (HAS_HOPS && r < bt) ? hops_rows[(size_t)r * n + jc] : 0;
    }
    // a strip column inside the block carries one diagonal entry: published from memory, untouched
    const bool in_blk = valid && j >= k0 && j < k0 + bt;
    const T dorig = in_blk ? rows[(size_t)(j - k0) * n + j] : T(0);
    const int32_t hdorig = (HAS_HOPS && in_blk) ? hops_rows[(size_t)(j - k0) * n + j] : 0;

    // one relaxation of row q by pivot t in both parts: cv = D_t[k0+r][k0+t], (wd, ws) = pivot row
    auto relax_row = [&](int q, T cv, int32_t ch, T wd, T ws, int32_t hwd, int32_t hws, int t) {
        if (MAXF) {
            pd[q] = fmax_t(pd[q], cv * wd);
            ps[q] = fmax_t(ps[q], cv * ws);
            return;
        }
        const T cd = cv * wd;
        const bool ud = pd[q] < cd;
        pd[q] = ud ? cd : pd[q];
        if (HAS_HOPS) hd[q] = ud ? (int32_t)((uint32_t)ch + (uint32_t)hwd) : hd[q];
        const T cs = cv * ws;
        const bool us = ps[q] < cs;
        ps[q] = us ? cs : ps[q];
        if (HAS_LAST) lp[q] = us ? k0 + t : lp[q];
        if (HAS_HOPS) hs[q] = us ? (int32_t)((uint32_t)ch + (uint32_t)hws) : hs[q];
    };

    // the serial phase of this wave's sub-block b: pivot by pivot, each line published (and, with flags, announced)
    // before the wave's other rows take the pivot
    auto serial_phase = [&](int b, int *flag) {
        T snap_v[SB];                                  // the snapshots: stored after the chain's four steps
        int32_t snap_l[SB], snap_h[SB];
#pragma unroll
        for (int tq = 0; tq < SB; ++tq) {
            const int t = b * SB + tq;
            if (t >= bt) continue;
            // the pivot row at time t: every earlier pivot has been applied, pivot t leaves it alone
            T wd = pd[tq];
            if (lane == t) wd = qnan<T>();     // skip j == k (also hides the stale diagonal)
            T ws = ps[tq];
            if (j == k0 + t) ws = dorig;
            const int32_t hwd = HAS_HOPS ? hd[tq] : 0;
            const int32_t hws = HAS_HOPS ? (j == k0 + t ? hdorig : hs[tq]) : 0;
            const T snap = ws;                                    // the snapshot (stored below)
            if (j == k0 + t) ws = qnan<T>();                      // skip j == k
            s_dline[t][lane] = wd;
            s_sline[t][lane] = ws;
            if (HAS_HOPS) { s_dh[t][lane] = hwd; s_sh[t][lane] = hws; }
            if (flag) panel_flag_raise(flag, t);
            snap_v[tq] = snap;
            snap_l[tq] = HAS_LAST ? (j == k0 + t ? -1 : lp[tq]) : 0;
            snap_h[tq] = hws;
            // column t of this wave's rows: SB independent cross-lane reads first
            T cv[SB];
            int32_t ch[SB];
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                cv[q] = readlane<T>(pd[q], t);
                ch[q] = HAS_HOPS ? readlane<int>(hd[q], t) : 0;
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                if (q == tq) continue;                        // skip i == k
                relax_row(q, cv[q], ch[q], wd, ws, hwd, hws, t);
            }
        }
        if (valid) {
#pragma unroll
            for (int tq = 0; tq < SB; ++tq) {
                const int t = b * SB + tq;
                if (t >= bt) continue;
                w_out[(size_t)t * n + j] = snap_v[tq];
                if (HAS_LAST) at_rows[(size_t)t * n + j] = snap_l[tq];
                if (HAS_HOPS) wh_out[(size_t)t * n + j] = snap_h[tq];
            }
        }
    };
    // published pivot t onto this wave's rows
    auto apply_pivot = [&](int t) {
        const T wd = s_dline[t][lane], ws = s_sline[t][lane];
        const int32_t hwd = HAS_HOPS ? s_dh[t][lane] : 0, hws = HAS_HOPS ? s_sh[t][lane] : 0;
        T cv[SB];
        int32_t ch[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
            cv[q] = readlane<T>(pd[q], t);
            ch[q] = HAS_HOPS ? readlane<int>(hd[q], t) : 0;
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) relax_row(q, cv[q], ch[q], wd, ws, hwd, hws, t);
    };
    if constexpr (FLAGS) {
    int *flag = reinterpret_cast<int *>(smem + rowpanel_lds<T, HAS_HOPS>() - PANEL_FLAG_BYTES);
    panel_flags_init(flag);
    __syncthreads();
    // sub-blocks long done: one wait each (the flag of their last pivot), four pivots per trip
#pragma unroll 1
    for (int b = 0; b + 1 < wave; ++b) {
        if (b * SB >= bt) break;                   // uniform
        panel_flag_wait(flag, min(b * SB + SB, bt) - 1);
#pragma unroll
        for (int tq = 0; tq < SB; ++tq)
            if (b * SB + tq < bt) apply_pivot(b * SB + tq);
    }
    // the sub-block right before this wave's: pivot by pivot, as its owner announces them
    if (wave > 0 && (wave - 1) * SB < bt) {
#pragma unroll
        for (int tq = 0; tq < SB; ++tq) {
            const int t = (wave - 1) * SB + tq;
            if (t >= bt) continue;
            panel_flag_wait(flag, t);
            apply_pivot(t);
        }
    }
    if (wave * SB < bt) serial_phase(wave, flag);
    } else {
#pragma unroll 1                                   // code size: keep the panel inside the I-cache
    for (int b = 0; b < B / SB; ++b) {
        if (b * SB >= bt) continue;                // uniform
        if (wave == b) serial_phase(b, nullptr);
        lds_barrier();
        if (wave > b) {            // earlier rows are past their own pivots: nothing needs them
#pragma unroll
            for (int tq = 0; tq < SB; ++tq)
                if (b * SB + tq < bt) apply_pivot(b * SB + tq);
        }
    }
    }
}

template <typename T, bool HAS_LAST, bool HAS_HOPS>
__global__ __launch_bounds__(PANEL_THREADS) void fused_rowpanel(const T *rows, int n, int k0, int bt,
                                                      T *w_out, const int32_t *last_rows,
                                                      int32_t *at_rows, const int32_t *hops_rows,
                                                      int32_t *wh_out)
{
    __shared__ __attribute__((aligned(16))) char smem[rowpanel_lds<T, HAS_HOPS>()];
    rowpanel_body<T, HAS_LAST, HAS_HOPS>(smem, (int)blockIdx.x, rows, n, k0, bt, w_out, last_rows, at_rows,
                                         hops_rows, wh_out);
}

// Column panel: time-t snapshots of the 64 pivot columns for 64 rows per workgroup (lanes = rows).
// Needs only the block columns of W (Wd[t][c] = W[t][k0+c]), so it runs on every rank.
// HAS_LAST: the path trace of the pivot columns (PathLog): `last` of each block entry is carried
// beside its rate, and at_col[i][k0+t] = last of (i, k0+t) at time k0+t is stored directly into the
// n x n matrix (single-GPU solves only: rows are global rows).
// HAS_HOPS: `hops` of the block columns are carried beside their rates (hops' = hops[i][k] +
// hops[k][j], the second operand from the hops panel wh of the pivot rows), and cht[t][i] = hops
// of (i, k0+t) at time k0+t is exported for the main kernel.
// OWN_D: the workgroup evolves the 64 x 64 diagonal block itself (from the pivot rows d_rows, their
// hops d_hops; lanes = block columns, exactly as fused_rowpanel does) TOGETHER with its column
// lines, instead of reading the block columns of a finished snapshot panel W: the column panel then
// does not depend on the row panel of its pass, and the two can share one launch (fused_panels).
// RW = rows per column workgroup: 64, or 32 -- half the lanes mirror the other half, twice the workgroups, and the
// lines of the published pivot columns are 32 wide: 32.25 KB of LDS instead of 48.25 (f32 with next-hops), which
// fits the hole ONE retiring workgroup of the 64 x 64 fused_main_arg leaves (LDS is handed out in one piece:
// profiles/r04_panels_32_rows.txt)
template <typename T, bool HAS_NEXT, bool HAS_HOPS, int RW = 64> constexpr int colpanel_lds()
{
    return B * RW * (int)sizeof(T) + (HAS_NEXT ? B * RW * 4 : 0) + B * B * (int)sizeof(T) +
           (HAS_HOPS ? B * RW * 4 + B * B * 4 : 0) + PANEL_FLAG_BYTES;
}
template <typename T, bool HAS_NEXT, bool HAS_LAST, bool HAS_HOPS, bool OWN_D, bool MAXF = false,
          bool FLAGS = FWX_PANEL_FLAGS != 0, int RW = 64>
__device__ __forceinline__ void colpanel_body(char *smem, int bid, const T *rate, const int32_t *next, int rows,
                                              int n, int row0, int k0, int bt, const T *w, T *ct, int32_t *cnt,
                                              int ct_ld, const int32_t *last, int32_t *at_col,
                                              const int32_t *hops, const int32_t *wh, int32_t *cht,
                                              const T *d_rows, const int32_t *d_hops)
{
    static_assert(RW == 64 || RW == 32, "rows per column workgroup");
    typedef T line_t[RW];                          // lines of the workgroup's rows
    typedef int32_t iline_t[RW];
    typedef T dline_t[64];                         // lines of the diagonal block
    typedef int32_t idline_t[64];
    __builtin_amdgcn_s_setprio(3);                 // the panels ARE the serial chain: first in line on their SIMD
    line_t *s_line = (line_t *)smem;               // published pivot columns (time-t, NaN at i==k)
    dline_t *s_wd = (dline_t *)(s_line + B);       // s_wd[t][c] = D_t[k0+t][k0+c]
    iline_t *s_nline = (iline_t *)(s_wd + B);      // (HAS_NEXT)
    iline_t *s_hline = s_nline + (HAS_NEXT ? B : 0);   // hops of the published pivot columns (HAS_HOPS)
    idline_t *s_wdh = (idline_t *)(s_hline + B);   // hops of D_t[k0+t][k0+c]
    static_assert(B == 64, "s_wd rows are 64 wide");
    const int lane = threadIdx.x & 63;
    const int rl = lane & (RW - 1);                // this lane's row of the workgroup (lanes RW.. mirror 0..RW-1)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int il = bid * RW + rl;
    const bool valid = il < rows && lane < RW;     // (the mirror lanes compute the same values and store nothing)
    const int ic = il < rows ? il : rows - 1;
    const int gi = row0 + ic;

    if (!OWN_D) {
        for (int idx = threadIdx.x; idx < B * B; idx += PANEL_THREADS) {
            const int t = idx / B, c = idx % B;
            s_wd[t][c] = (t < bt && c < bt) ? w[(size_t)t * n + k0 + c] : qnan<T>();
            if (HAS_HOPS) s_wdh[t][c] = (t < bt && c < bt) ? wh[(size_t)t * n + k0 + c] : 0;
        }
    }

    T d[SB];
    int32_t nx[SB], lp[HAS_LAST ? SB : 1], hd[SB];
    T pd[OWN_D ? SB : 1];                          // this wave's SB rows of the diagonal block
    int32_t hdd[OWN_D ? SB : 1];                   // their hops
#pragma unroll
    for (int q = 0; q < SB; ++q) {
        const int c = wave * SB + q;
        const size_t off = (size_t)ic * n + k0 + (c < bt ? c : 0);
        d[q] = c < bt ? rate[off] : qnan<T>();
        nx[q] = (HAS_NEXT && c < bt) ? next[off] : -1;
        if (HAS_LAST) lp[q] = c < bt ? last[off] : -1;
        hd[q] = (HAS_HOPS && c < bt) ? hops[off] : 0;
        if (OWN_D) {
            pd[q] = (c < bt && lane < bt) ? d_rows[(size_t)c * n + k0 + lane] : qnan<T>();
            hdd[q] = (HAS_HOPS && c < bt && lane < bt) ? d_hops[(size_t)c * n + k0 + lane] : 0;
        }
    }
    if constexpr (FLAGS)
        panel_flags_init(reinterpret_cast<int *>(smem + colpanel_lds<T, HAS_NEXT, HAS_HOPS, RW>() - PANEL_FLAG_BYTES));
    __syncthreads();

    // the serial phase of this wave's sub-block b (see rowpanel_body)
    auto serial_phase = [&](int b, int *flag) {
        T snap_c[SB];                                  // the snapshots: stored after the chain's four steps
        int32_t snap_n[SB], snap_l[SB], snap_h[SB];
#pragma unroll
        for (int tq = 0; tq < SB; ++tq) {
            const int t = b * SB + tq;
            if (t >= bt) continue;
            // ---- the diagonal block: pivot row t at time t (NaN at its own column: skip j == k,
            // which also hides the stale diagonal entry), then this wave's other rows
            T wd = T(0);
            int32_t hwd = 0;
            if (OWN_D) {
                wd = pd[tq];
                if (lane == t) wd = qnan<T>();
                hwd = HAS_HOPS ? hdd[tq] : 0;
                s_wd[t][lane] = wd;
                if (HAS_HOPS) s_wdh[t][lane] = hwd;
            }
            // ---- the column lines
            T c = d[tq];
            const int32_t cn = nx[tq];
            const int32_t hc = hd[tq];
            if (gi == k0 + t) c = qnan<T>();                  // skip i == k
            s_line[t][rl] = c;
            if (HAS_NEXT) s_nline[t][rl] = cn;
            if (HAS_HOPS) s_hline[t][rl] = hc;
            if (flag) panel_flag_raise(flag, t);
            snap_c[tq] = c;
            snap_n[tq] = cn;
            snap_l[tq] = HAS_LAST ? (gi == k0 + t ? -1 : lp[tq]) : 0;
            snap_h[tq] = hc;
            if (OWN_D) {
                T cv[SB];
                int32_t chv[SB];
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    cv[q] = readlane<T>(pd[q], t);
                    chv[q] = HAS_HOPS ? readlane<int>(hdd[q], t) : 0;
                }
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    if (q == tq) continue;                    // skip i == k
                    const T cd = cv[q] * wd;
                    if (MAXF) { pd[q] = fmax_t(pd[q], cd); continue; }
                    const bool ud = pd[q] < cd;
                    pd[q] = ud ? cd : pd[q];
                    if (HAS_HOPS) hdd[q] = ud ? (int32_t)((uint32_t)chv[q] + (uint32_t)hwd) : hdd[q];
                }
            }
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                if (q == tq) continue;                        // skip j == k
                // D_t[k0+t][k0 + b*SB + q]: lane b*SB+q of the pivot row just published
                const T wv = OWN_D ? readlane<T>(wd, b * SB + q) : s_wd[t][b * SB + q];
                const int32_t wvh = !HAS_HOPS ? 0 : OWN_D ? readlane<int>(hwd, b * SB + q) : s_wdh[t][b * SB + q];
                const T cand = c * wv;
                if (MAXF) { d[q] = fmax_t(d[q], cand); continue; }
                const bool up = d[q] < cand;
                d[q] = up ? cand : d[q];
                if (HAS_NEXT) nx[q] = up ? cn : nx[q];
                if (HAS_LAST) lp[q] = up ? k0 + t : lp[q];
                if (HAS_HOPS) hd[q] = up ? (int32_t)((uint32_t)hc + (uint32_t)wvh) : hd[q];
            }
        }
        if (valid) {
#pragma unroll
            for (int tq = 0; tq < SB; ++tq) {
                const int t = b * SB + tq;
                if (t >= bt) continue;
                ct[(size_t)t * ct_ld + il] = snap_c[tq];
                if (HAS_NEXT) cnt[(size_t)t * ct_ld + il] = snap_n[tq];
                if (HAS_LAST) at_col[(size_t)il * n + k0 + t] = snap_l[tq];
                if (HAS_HOPS) cht[(size_t)t * ct_ld + il] = snap_h[tq];
            }
        }
    };
    // published pivot t onto this wave's lines (columns left of the sub-block have been snapshotted already; rows
    // above it are past their own pivots too)
    auto apply_pivot = [&](int t) {
        if (OWN_D) {
            const T wd = s_wd[t][lane];
            const int32_t hwd = HAS_HOPS ? s_wdh[t][lane] : 0;
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const T cd = readlane<T>(pd[q], t) * wd;
                if (MAXF) { pd[q] = fmax_t(pd[q], cd); continue; }
                const int32_t chv = HAS_HOPS ? readlane<int>(hdd[q], t) : 0;
                const bool ud = pd[q] < cd;
                pd[q] = ud ? cd : pd[q];
                if (HAS_HOPS) hdd[q] = ud ? (int32_t)((uint32_t)chv + (uint32_t)hwd) : hdd[q];
            }
        }
        const T c = s_line[t][rl];
        const int32_t cn = HAS_NEXT ? s_nline[t][rl] : 0;
        const int32_t hc = HAS_HOPS ? s_hline[t][rl] : 0;
#pragma unroll
        for (int q = 0; q < SB; ++q) {
            const T cand = c * s_wd[t][wave * SB + q];
            if (MAXF) { d[q] = fmax_t(d[q], cand); continue; }
            const bool up = d[q] < cand;
            d[q] = up ? cand : d[q];
            if (HAS_NEXT) nx[q] = up ? cn : nx[q];
            if (HAS_LAST) lp[q] = up ? k0 + t : lp[q];
            if (HAS_HOPS) hd[q] = up ? (int32_t)((uint32_t)hc + (uint32_t)s_wdh[t][wave * SB + q]) : hd[q];
        }
    };
    if constexpr (FLAGS) {
    int *flag = reinterpret_cast<int *>(smem + colpanel_lds<T, HAS_NEXT, HAS_HOPS, RW>() - PANEL_FLAG_BYTES);
    // (the flags were initialised before the barrier above)
#pragma unroll 1
    for (int b = 0; b + 1 < wave; ++b) {
        if (b * SB >= bt) break;                   // uniform
        panel_flag_wait(flag, min(b * SB + SB, bt) - 1);
#pragma unroll
        for (int tq = 0; tq < SB; ++tq)
            if (b * SB + tq < bt) apply_pivot(b * SB + tq);
    }
    if (wave > 0 && (wave - 1) * SB < bt) {
#pragma unroll
        for (int tq = 0; tq < SB; ++tq) {
            const int t = (wave - 1) * SB + tq;
            if (t >= bt) continue;
            panel_flag_wait(flag, t);
            apply_pivot(t);
        }
    }
    if (wave * SB < bt) serial_phase(wave, flag);
    } else {
#pragma unroll 1
    for (int b = 0; b < B / SB; ++b) {
        if (b * SB >= bt) continue;
        if (wave == b) serial_phase(b, nullptr);
        lds_barrier();
        if (wave > b) {
#pragma unroll
            for (int tq = 0; tq < SB; ++tq)
                if (b * SB + tq < bt) apply_pivot(b * SB + tq);
        }
    }
    }
}

template <typename T, bool HAS_NEXT, bool HAS_LAST, bool HAS_HOPS>
__global__ __launch_bounds__(PANEL_THREADS) void fused_colpanel(const T *rate, const int32_t *next, int rows,
                                                      int n, int row0, int k0, int bt, const T *w,
                                                      T *ct, int32_t *cnt, int ct_ld,
                                                      const int32_t *last, int32_t *at_col,
                                                      const int32_t *hops, const int32_t *wh,
                                                      int32_t *cht)
{
    __shared__ __attribute__((aligned(16))) char smem[colpanel_lds<T, HAS_NEXT, HAS_HOPS>()];
    colpanel_body<T, HAS_NEXT, HAS_LAST, HAS_HOPS, false>(smem, (int)blockIdx.x, rate, next, rows, n, row0, k0, bt,
                                                          w, ct, cnt, ct_ld, last, at_col, hops, wh, cht,
                                                          nullptr, nullptr);
}

// Both panels of a pass in ONE launch (single-device solves: the slab is the whole matrix): the
// first n/64 workgroups run the row panel, the rest the column panel with its own diagonal block.
template <typename T, bool HAS_NEXT, bool HAS_LAST, bool HAS_HOPS, bool MAXF = false>
__global__ __launch_bounds__(PANEL_THREADS) void fused_panels(int row_wgs, const T *rate, const int32_t *next,
                                                    int n, int k0, int bt, T *w_out, T *ct, int32_t *cnt,
                                                    int ct_ld, const int32_t *last, int32_t *at_row,
                                                    int32_t *at_col, const int32_t *hops, int32_t *wh_out,
                                                    int32_t *cht)
{
    constexpr int RL = rowpanel_lds<T, HAS_HOPS>(), CL = colpanel_lds<T, HAS_NEXT, HAS_HOPS>();
    __shared__ __attribute__((aligned(16))) char smem[RL > CL ? RL : CL];
    // flags everywhere but the f32 max-form (rates only, inside the domain) panels: their pivots are so cheap that
    // the polling costs more than the overlap gives (N = 1024: 0.47 -> 0.48 ms; with next-hops 0.68 -> 0.61)
    constexpr bool FL = FWX_PANEL_FLAGS != 0 && !(MAXF && sizeof(T) == 4);
    const size_t prow = (size_t)k0 * n;            // the pivot rows
    if ((int)blockIdx.x < row_wgs) {               // workgroup-uniform
        rowpanel_body<T, HAS_LAST, HAS_HOPS, MAXF, FL>(smem, (int)blockIdx.x, rate + prow, n, k0, bt, w_out,
                                             HAS_LAST ? last + prow : nullptr, HAS_LAST ? at_row + prow : nullptr,
                                             HAS_HOPS ? hops + prow : nullptr, wh_out);
    } else {
        colpanel_body<T, HAS_NEXT, HAS_LAST, HAS_HOPS, true, MAXF, FL>(smem, (int)blockIdx.x - row_wgs, rate, next, n, n, 0,
                                                             k0, bt, nullptr, ct, cnt, ct_ld, last, at_col, hops,
                                                             nullptr, cht, rate + prow,
                                                             HAS_HOPS ? hops + prow : nullptr);
    }
}

// The same for f32 rates + next-hops, with or without the path trace (no hops), held to 48 VGPRs.  A
// 1024-thread workgroup is four waves per SIMD: at 48 registers they take 192, and a retiring
// fused_main_arg workgroup (three per CU at 152 allocated registers per wave) leaves 512 - 2 * 152 = 208
// free -- so this launch's workgroups are placed one by one as main workgroups retire.  At the 49 / 54
// (allocated: 56) the compiler picks by itself they need 224: TWO free main slots on one CU, i.e. the
// main grid's tail (772 us beside the main launch against 30 us alone, gpurun_out/r03_final_prof_next),
// and with two passes per main launch the side chain -- two panels per pair -- then ends after it.
// (amdgpu_num_vgpr counts HALF the unified register file of gfx950: 24 -> 48.)
// Since round 4 (flags instead of barriers, snapshot stores after the serial phase) the bodies need 32 / 36
// registers with or without this cap; the kernel stays as the place where the budget is ENFORCED, and as the
// carrier of RW = 32: column workgroups of 32 rows (see colpanel_lds), twice as many -- 32.25 KB of LDS and 32
// registers per workgroup, which fits the hole one retiring workgroup of the 64 x 64 fused_main_arg leaves.
template <bool HAS_LAST, int RW = 64>
__global__ __launch_bounds__(PANEL_THREADS) __attribute__((amdgpu_num_vgpr(24)))
void fused_panels_next_f32(int row_wgs, const float *rate, const int32_t *next, int n, int k0, int bt, float *w_out,
                           float *ct, int32_t *cnt, int ct_ld, const int32_t *last, int32_t *at_row,
                           int32_t *at_col)
{
    constexpr int RL = rowpanel_lds<float, false>(), CL = colpanel_lds<float, true, false, RW>();
    __shared__ __attribute__((aligned(16))) char smem[RL > CL ? RL : CL];
    const size_t prow = (size_t)k0 * n;
    if ((int)blockIdx.x < row_wgs)                 // workgroup-uniform
        rowpanel_body<float, HAS_LAST, false, false>(smem, (int)blockIdx.x, rate + prow, n, k0, bt, w_out,
                                                     HAS_LAST ? last + prow : nullptr,
                                                     HAS_LAST ? at_row + prow : nullptr, nullptr, nullptr);
    else
        colpanel_body<float, true, HAS_LAST, false, true, false, FWX_PANEL_FLAGS != 0, RW>(
            smem, (int)blockIdx.x - row_wgs, rate, next, n, n, 0, k0, bt, nullptr, ct, cnt, ct_ld, last, at_col,
            nullptr, nullptr, nullptr, rate + prow, nullptr);
}

// ------------------------------------------------------------------------------------------------
// BS pivots are staged in LDS at a time (the register tile lives across stages), which keeps a
// workgroup at 32 KiB of LDS and <= 128 VGPRs: 4 workgroups = 16 waves per CU.  The loop is pure
// VALU (v_pk_mul_f32 + v_cmp + v_cndmask per pair of relaxations) and one wave alone issues at
// half rate on a SIMD-32, so occupancy, not bytes, is what this kernel needs.
// TRACK: per entry, the pivot of its newest update in this pass is kept in a register tile; the
// write-back uses it for the path trace (last != nullptr) and for hops (hops != nullptr:
// hops = cht[t*][i] + wh[t*][j], the lengths of the two halves Algorithms.hs:55 concatenates, as
// exported by the column / row panels).
template <typename T, bool HAS_NEXT, bool COUNT, int BS, int MINW, int NH, int RI, bool MAXF = false,
          bool TRACK = false>
__global__ __launch_bounds__(256, MINW) void fused_main(T *rate, int32_t *next, int rows, int n,
                                                        int row0, int k0, int bt, const T *w,
                                                        const T *ct, const int32_t *cnt, int ct_ld,
                                                        int skip_lo, int skip_hi,
                                                        unsigned long long *updates, int32_t *last,
                                                        int32_t *hops, const int32_t *cht,
                                                        const int32_t *wh, ColWin cw)
{
    if (cw.prio) __builtin_amdgcn_s_setprio(2);    // a launch of the look-ahead chain (FusedArgs::side)
    constexpr bool HAS_LAST = TRACK;
    static_assert(!HAS_LAST || HAS_NEXT, "the path trace and hops ride on the next-hop variant");
    using V = typename Vec16<T>::type;
    using IV = typename IVec<Vec16<T>::W>::type;
    constexpr int VW = Vec16<T>::W;
    constexpr int TI = 16 * RI;               // RI = 8: 128 rows; RI = 4: 64 (small matrices)
    constexpr int TJ = 16 * NH * VW;          // NH = 2: 128 (f32) / 64 (f64) columns; NH = 1: half
    constexpr int HJ = TJ / NH;               // NH 16-byte vectors per thread and row

    __shared__ __attribute__((aligned(16))) T sW[BS][TJ];
    __shared__ __attribute__((aligned(16))) T sC[BS][TI];
    __shared__ __attribute__((aligned(16))) int32_t sN[HAS_NEXT ? BS : 1][HAS_NEXT ? TI : 4];
    __shared__ unsigned int s_cnt;

    const int tid = threadIdx.x;
    const int i_base = blockIdx.y * TI;
    const int j_base = (blockIdx.x + cw.jt0) * TJ;
    if (COUNT && tid == 0) s_cnt = 0;

    // ---- this thread's 8 x (2 vectors) register tile -------------------------------------------
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    // rows [skip_lo, skip_hi) (multiples of 8 = one thread's rows) were relaxed by an earlier
    // launch of this pass (look-ahead): their threads only help with staging and barriers
    const bool skip = i0 >= skip_lo && i0 < skip_hi;
    int jcol[NH];
    bool jok[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int j = j_base + h * HJ + tj * VW;
        jok[h] = j < n && !cw.skips(j);
        jcol[h] = jok[h] ? j : n - VW;
    }
    V x[RI][NH];
    IV nx[HAS_NEXT ? RI : 1][NH];
    IV lp[HAS_LAST ? RI : 1][NH];    // path trace: pivot of the newest update in this pass, or -2
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = min(i0 + r, rows - 1);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            x[r][h] = *reinterpret_cast<const V *>(rate + (size_t)i * n + jcol[h]);
            // the next-hop tile is NOT read: -2 marks "unchanged" (real values are >= -1) and
            // only vectors with a changed component are read-modified-written at the end
            if (HAS_NEXT)
#pragma unroll
                for (int e = 0; e < VW; ++e) nx[r][h][e] = -2;
            if (HAS_LAST)
#pragma unroll
                for (int e = 0; e < VW; ++e) lp[r][h][e] = -2;
        }
    }

    // A diagonal tile holds entries with i == j, which no step may touch (Algorithms.hs:54).
    const int gi_lo = row0 + i_base, gj_lo = j_base;
    const bool diag_tile = gi_lo < gj_lo + TJ && gj_lo < gi_lo + TI;
    unsigned int my_updates = 0;

    for (int s0 = 0; s0 < bt; s0 += BS) {
        const int bs = min(BS, bt - s0);
        if (s0) __syncthreads();              // everyone is done reading the previous stage
        // ---- stage W (NaN at j == k: skip j==k) and C for pivots [s0, s0+bs) -------------------
        constexpr int WV_PER_ROW = TJ / VW;
        for (int idx = tid; idx < BS * WV_PER_ROW; idx += 256) {
            const int tl = idx / WV_PER_ROW, jv = idx % WV_PER_ROW;
            const int t = s0 + tl;
            const int j = j_base + jv * VW;
            V v;
            if (tl < bs && j < n) {
                v = *reinterpret_cast<const V *>(w + (size_t)t * n + j);
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    if (j + e == k0 + t) v[e] = qnan<T>();
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) v[e] = qnan<T>();
            }
            *reinterpret_cast<V *>(&sW[tl][jv * VW]) = v;
        }
        for (int idx = tid; idx < BS * TI; idx += 256) {
            const int tl = idx / TI, il = idx % TI;
            const int i = i_base + il;
            const bool ok = tl < bs && i < rows;
            sC[tl][il] = ok ? ct[(size_t)(s0 + tl) * ct_ld + i] : qnan<T>();
            if (HAS_NEXT) sN[tl][il] = ok ? cnt[(size_t)(s0 + tl) * ct_ld + i] : -1;
        }
        __syncthreads();

        // ---- bs in-order relaxations per entry, operands from LDS ------------------------------
        for (int tl = 0; tl < (skip ? 0 : bs); ++tl) {
            T c[RI];
            int32_t cn[HAS_NEXT ? RI : 1];
            V wv[NH];
#pragma unroll
            for (int q = 0; q < RI / VW; ++q) {
                const V cv = *reinterpret_cast<const V *>(&sC[tl][ti * RI + q * VW]);
#pragma unroll
                for (int e = 0; e < VW; ++e) c[q * VW + e] = cv[e];
                if (HAS_NEXT) {
                    const IV nv = *reinterpret_cast<const IV *>(&sN[tl][ti * RI + q * VW]);
#pragma unroll
                    for (int e = 0; e < VW; ++e) cn[q * VW + e] = nv[e];
                }
            }
#pragma unroll
            for (int h = 0; h < NH; ++h)
                wv[h] = *reinterpret_cast<const V *>(&sW[tl][h * HJ + tj * VW]);
#pragma unroll
            for (int r = 0; r < RI; ++r)
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        const T cand = c[r] * wv[h][e];
                        if (MAXF) {   // max-form domain (see fused_main_max): strict fold == max
                            x[r][h][e] = fmax_t(x[r][h][e], cand);
                            continue;
                        }
                        const bool up = x[r][h][e] < cand;
                        if (COUNT) {
                            const bool is_diag = diag_tile && (row0 + i0 + r == jcol[h] + e);
                            my_updates += (up && !is_diag && jok[h] && i0 + r < rows) ? 1u : 0u;
                        }
                        x[r][h][e] = up ? cand : x[r][h][e];
                        if (HAS_NEXT) nx[r][h][e] = up ? cn[r] : nx[r][h][e];
                        if (HAS_LAST) lp[r][h][e] = up ? k0 + s0 + tl : lp[r][h][e];
                    }
        }
    }

    // ---- write back (the diagonal entry, if any, is restored from memory first: once, with its
    // own wait -- a conditional load inside the store loop would make every store wait for all
    // the stores before it, vmcnt counts in order) ----------------------------------------------
    if (diag_tile) {                                 // workgroup-uniform
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int i = i0 + r, gi = row0 + i;
            if (i >= rows || skip) continue;
#pragma unroll
            for (int h = 0; h < NH; ++h)
                if (jok[h] && gi >= jcol[h] && gi < jcol[h] + VW) {
                    const int e = gi - jcol[h];
                    x[r][h][e] = rate[(size_t)i * n + gi];
                    if (HAS_NEXT) nx[r][h][e] = -2;
                    if (HAS_LAST) lp[r][h][e] = -2;
                }
        }
        __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
    }
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = i0 + r;
        if (i >= rows || skip) continue;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (!jok[h]) continue;
            const size_t off = (size_t)i * n + jcol[h];
            if (HAS_NEXT) {
                // a rate changed <=> its next-hop was set: unchanged vectors are not written at all
                bool any = false;
#pragma unroll
                for (int e = 0; e < VW; ++e) any |= nx[r][h][e] != -2;
                if (any) {
                    // changed components only, as plain 4-byte stores: a read-modify-write of the
                    // vector would put a load and its wait (vmcnt counts in order: a wait for every
                    // store issued so far as well) in front of each of the tile's vectors
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        if (nx[r][h][e] != -2) next[off + e] = nx[r][h][e];
                    *reinterpret_cast<V *>(rate + off) = x[r][h];
                    if (HAS_LAST && hops) {   // lengths of the two halves at the winning pivot
#pragma unroll
                        for (int e = 0; e < VW; ++e)
                            if (lp[r][h][e] != -2) {
                                const int t = lp[r][h][e] - k0;
                                hops[off + e] = (int32_t)((uint32_t)cht[(size_t)t * ct_ld + i] +
                                                          (uint32_t)wh[(size_t)t * n + jcol[h] + e]);
                            }
                    }
                    if (HAS_LAST && last) {   // same vectors, same components: changed <=> lp != -2
#pragma unroll
                        for (int e = 0; e < VW; ++e)
                            if (lp[r][h][e] != -2) last[off + e] = lp[r][h][e];
                    }
                }
            } else {
                *reinterpret_cast<V *>(rate + off) = x[r][h];
            }
        }
    }

    if (COUNT) {
        if (my_updates) atomicAdd(&s_cnt, my_updates);
        __syncthreads();
        if (tid == 0 && s_cnt)
            atomicAdd(&updates[(blockIdx.x + blockIdx.y * 7) & (FWX_UPDATE_SHARDS_K - 1)],
                      (unsigned long long)s_cnt);
    }
}

typedef float F32x4 __attribute__((ext_vector_type(4)));

// One pivot pair of the max-form fold on a thread's RI x (NH x 4) register tile: operands from the
// LDS stage ([u]: pivot u of the pair), all 16 products of a row first, then its 8 folds, so a
// v_max3 never issues right behind the multiplies it depends on.  Two plain v_mul_f32 rather than
// one v_pk_mul_f32: same issue cycles per pair, no register-pair shuffles (measured).
template <int RI, int NH>
__device__ __forceinline__ void max_fold_pair(const float (&sWp)[2][64 * NH], const float (&sCp)[2][16 * RI],
                                              int ti, int tj, F32x4 (&x)[RI][NH])
{
    float c[RI][2], wv[NH][4][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int q = 0; q < RI / 4; ++q) {
            const F32x4 cv = *reinterpret_cast<const F32x4 *>(&sCp[u][ti * RI + q * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) c[q * 4 + e][u] = cv[e];
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const F32x4 wq = *reinterpret_cast<const F32x4 *>(&sWp[u][h * 64 + tj * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[h][e][u] = wq[e];
        }
    }
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        float p0[NH][4], p1[NH][4];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm("v_mul_f32 %0, %1, %2" : "=v"(p0[h][e]) : "v"(c[r][0]), "v"(wv[h][e][0]));
                asm("v_mul_f32 %0, %1, %2" : "=v"(p1[h][e]) : "v"(c[r][1]), "v"(wv[h][e][1]));
            }
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                x[r][h][e] = __builtin_fmaxf(__builtin_fmaxf(x[r][h][e], p0[h][e]), p1[h][e]);
    }
}

// Interior tiles of a full pass (the tile lies inside the slab, off the diagonal, away from the
// pivot columns and from the look-ahead rows; bt == B): nothing to bound-check, nothing to patch,
// nothing to restore.  At N = 16384 that is 98 % of the tiles, and the general path executes about
// 900 instructions of address arithmetic, predicates and branches per tile and wave around the
// 6144 of the fold -- scalar and 64-bit VALU work that issues at 4+ cycles, i.e. a fifth of the
// kernel (tools/experiments/gen_loop_replay.py: the fold loop alone runs at 1.17 ns per
// instruction, which would be 460 us per launch; the launch took 618).
// ns: 16-pivot stages per tile = pivots of the launch / 16 (4: one pass; 8: a double pass, see
// fused_range); even.  The stage loop is a RUNTIME loop over pairs of stages (buffer 0, buffer 1):
// one copy of the code for both pass lengths -- two unrolled instantiations in one kernel cost
// 13 VGPRs and pushed the tile into scratch.
template <int RI, int NH>
__device__ __forceinline__ void main_max_interior(float *rate, int n, int i_base, int j_base, const float *w,
                                                  const float *ct, int ct_ld,
                                                  float (&sW)[2][8][2][64 * NH], float (&sC)[2][8][2][16 * RI],
                                                  int ns)
{
    constexpr int TI = 16 * RI, TJ = 64 * NH, BS = 16;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int sp = tid >> 5, sv = tid & 31;
    const bool sw_role = sv * 4 < TJ, sc_role = sv * 4 < TI;
    const float *wp = w + (size_t)(2 * sp) * n + j_base + sv * 4;
    const float *cp = ct + (size_t)(2 * sp) * ct_ld + i_base + sv * 4;
    F32x4 pw[2], pc[2];
    auto prefetch = [&]() {
        if (sw_role) {
            pw[0] = *reinterpret_cast<const F32x4 *>(wp);
            pw[1] = *reinterpret_cast<const F32x4 *>(wp + n);
        }
        if (sc_role) {
            pc[0] = *reinterpret_cast<const F32x4 *>(cp);
            pc[1] = *reinterpret_cast<const F32x4 *>(cp + ct_ld);
        }
        wp += (size_t)BS * n;
        cp += (size_t)BS * ct_ld;
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (sw_role) *reinterpret_cast<F32x4 *>(&sW[buf][sp][u][sv * 4]) = pw[u];
            if (sc_role) *reinterpret_cast<F32x4 *>(&sC[buf][sp][u][sv * 4]) = pc[u];
        }
    };
    prefetch();
    float *xp = rate + (size_t)(i_base + ti * RI) * n + j_base + tj * 4;
    F32x4 x[RI][NH];
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) x[r][h] = *reinterpret_cast<const F32x4 *>(xp + (size_t)r * n + h * 64);
    commit(0);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);   // the tile is complete here (see fused_main_max)
#pragma unroll 1
    for (int s = 0; s < ns; s += 2) {
        prefetch();                                  // stage s + 1: in flight during the fold below
#pragma unroll 1
        for (int tp = 0; tp < BS / 2; ++tp) max_fold_pair<RI, NH>(sW[0][tp], sC[0][tp], ti, tj, x);
        commit(1);
        __syncthreads();
        const bool more = s + 2 < ns;                // workgroup-uniform
        if (more) prefetch();
#pragma unroll 1
        for (int tp = 0; tp < BS / 2; ++tp) max_fold_pair<RI, NH>(sW[1][tp], sC[1][tp], ti, tj, x);
        if (more) {
            commit(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int h = 0; h < NH; ++h) *reinterpret_cast<F32x4 *>(xp + (size_t)r * n + h * 64) = x[r][h];
}

// ------------------------------------------------------------------------------------------------
// fused_main_max: rates-only f32 main kernel for matrices whose entries are all >= +0 and not NaN
// (what the reference's parser guarantees: rates > 0, Parsers.hs:40; unreachable = +0.0).
//
// On that domain the strict fold  x <- (x < c) ? c : x  equals  x <- max(x, c)  BIT FOR BIT:
// every candidate is a product of non-negative operands, hence >= +0, +inf, or NaN (inf * 0);
// x is never NaN and never -0 (it only ever takes the value of a candidate that won a strict
// compare), max(x, NaN) = x exactly as `x < NaN` is false, and for x == c both forms leave the
// same bits.  max is associative, so two pivots fold per instruction:
//       x <- max3(x, C_t[i]*W_t[j], C_{t+1}[i]*W_{t+1}[j])
// with the operands stored as (t, t+1) pairs in LDS: 8.0 issue cycles per pair of relaxations
// (two 2-cycle v_mul_f32 + one 4-cycle v_max3_f32; tools/valu_rate.hip) instead of 24 for the
// compare form.  The caller must have verified the domain (fwx_dev_check_nonneg); with next-hops
// the same fold runs in fused_main_arg, which recovers the winning pivot afterwards.
// ------------------------------------------------------------------------------------------------
template <int MINW, int UNR, int RI, int NH>
__global__ __launch_bounds__(256, MINW) void fused_main_max(float *rate, int rows, int n, int row0,
                                                            int k0, int bt, const float *w,
                                                            const float *ct, int ct_ld, int ct_vec,
                                                            int skip_lo, int skip_hi, ColWin cw)
{
    FWX_PROBE;
    if (cw.prio) __builtin_amdgcn_s_setprio(2);    // a launch of the look-ahead chain (FusedArgs::side)
    typedef float V4 __attribute__((ext_vector_type(4)));
    // 16 pivots (8 pairs) per LDS stage, two stages resident: while stage s is being folded the
    // operands of stage s+1 are already in flight from L2 into registers.
    constexpr int TI = 16 * RI, TJ = 64 * NH, HJ = 64, BS = 16, HP = BS / 2;

    // [buffer][pivot pair][even/odd pivot of the pair][column or row]: a thread's 16-byte reads of
    // W are 16 bytes apart across the 16 lanes of a read group -> conflict-free
    __shared__ __attribute__((aligned(16))) float sW[2][HP][2][TJ];
    __shared__ __attribute__((aligned(16))) float sC[2][HP][2][TI];

    const int tid = threadIdx.x;
    const int i_base = blockIdx.y * TI;
    const int j_base = (blockIdx.x + cw.jt0) * TJ;
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    {
        const int gi = row0 + i_base;
        const bool interior = (bt == B || bt == 2 * B) && ct_vec && i_base + TI <= rows && j_base + TJ <= n &&
                              !(gi < j_base + TJ && j_base < gi + TI) &&          // off the diagonal
                              (k0 + bt <= j_base || k0 >= j_base + TJ) &&        // no pivot column
                              (i_base + TI <= skip_lo || i_base >= skip_hi) &&   // no look-ahead rows
                              cw.clear_of(j_base, j_base + TJ);                  // ... or columns
        if (interior) {                                                          // workgroup-uniform
            main_max_interior<RI, NH>(rate, n, i_base, j_base, w, ct, ct_ld, sW, sC, bt / 16);
            return;
        }
    }
    const bool skip = i0 >= skip_lo && i0 < skip_hi;   // rows done by the look-ahead launch
    const float nanv = qnan<float>();

    // staging role of this thread: pivot pair sp (0..7), 4 consecutive columns / rows at sv*4
    const int sp = tid >> 5, sv = tid & 31;
    const int sj = j_base + sv * 4;              // W columns
    const int si = i_base + sv * 4;              // C rows
    const bool sw_role = sv * 4 < TJ;            // small tiles: only some threads stage
    const bool sc_role = sv * 4 < TI;
    const bool sj_ok = sj < n && sw_role;        // n % 4 == 0: whole vector in or out
    V4 pw[2], pc[2];
    auto prefetch = [&](int s0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = s0 + 2 * sp + u;
            const bool t_ok = t < bt;
            pw[u] = (t_ok && sj_ok) ? *reinterpret_cast<const V4 *>(w + (size_t)t * n + sj)
                                    : V4{nanv, nanv, nanv, nanv};
            if (!sc_role) {
                pc[u] = V4{nanv, nanv, nanv, nanv};
            } else if (t_ok && ct_vec && si + 4 <= rows) {
                pc[u] = *reinterpret_cast<const V4 *>(ct + (size_t)t * ct_ld + si);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pc[u][e] = (t_ok && si + e < rows) ? ct[(size_t)t * ct_ld + si + e] : nanv;
            }
            // skip j == k: the pivot's own column
            const int kcol = k0 + t - sj;
            if (kcol >= 0 && kcol < 4) pw[u][kcol] = nanv;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (sw_role) *reinterpret_cast<V4 *>(&sW[buf][sp][u][sv * 4]) = pw[u];
            if (sc_role) *reinterpret_cast<V4 *>(&sC[buf][sp][u][sv * 4]) = pc[u];
        }
    };

    prefetch(0);

    int jcol[NH];
    bool jok[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int j = j_base + h * HJ + tj * 4;
        jok[h] = j < n && !cw.skips(j);
        jcol[h] = jok[h] ? j : n - 4;
    }
    V4 x[RI][NH];
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = min(i0 + r, rows - 1);
#pragma unroll
        for (int h = 0; h < NH; ++h)
            x[r][h] = *reinterpret_cast<const V4 *>(rate + (size_t)i * n + jcol[h]);
    }
    const int gi_lo = row0 + i_base, gj_lo = j_base;
    const bool diag_tile = gi_lo < gj_lo + TJ && gj_lo < gi_lo + TI;

    commit(0);
    __syncthreads();
    // The tile is complete HERE, once.  Without this the compiler's wait-count pass carries "x may
    // still be loading" into the fold loop and guards every first use of a tile register with
    // s_waitcnt vmcnt(k), k = 15..0 -- and vmcnt counts the stage-(s+1) operand loads issued just
    // above the fold as well, so each stage stalled on its own prefetch.
    __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);

    int buf = 0;
    for (int s0 = 0; s0 < bt; s0 += BS, buf ^= 1) {
        const bool more = s0 + BS < bt;
        if (more) prefetch(s0 + BS);             // in flight during the fold below
        const int np = skip ? 0 : (min(BS, bt - s0) + 1) / 2;
#pragma unroll UNR
        for (int tp = 0; tp < np; ++tp) max_fold_pair<RI, NH>(sW[buf][tp], sC[buf][tp], ti, tj, x);
        if (more) {
            commit(buf ^ 1);                      // the other buffer: nobody reads it now
            __syncthreads();
        }
    }

    if (diag_tile) {        // the diagonal entries keep their value: restored once, with one wait
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int i = i0 + r, gi = row0 + i;
            if (i >= rows || skip) continue;
#pragma unroll
            for (int h = 0; h < NH; ++h)
                if (jok[h] && gi >= jcol[h] && gi < jcol[h] + 4) x[r][h][gi - jcol[h]] = rate[(size_t)i * n + gi];
        }
        __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
    }
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = i0 + r;
        if (i >= rows || skip) continue;
#pragma unroll
        for (int h = 0; h < NH; ++h)
            if (jok[h]) *reinterpret_cast<V4 *>(rate + (size_t)i * n + jcol[h]) = x[r][h];
    }
}

typedef double F64x2 __attribute__((ext_vector_type(2)));

// One pivot of the f64 max-form fold on a thread's 8 x (4 x 2) register tile: the 8 products of a
// row first, then its 8 folds.
__device__ __forceinline__ void max_fold_f64(const double (&sWt)[128], const double (&sCt)[128], int ti, int tj,
                                             F64x2 (&x)[8][4])
{
    double c[8], wv[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const F64x2 cv = *reinterpret_cast<const F64x2 *>(&sCt[ti * 8 + q * 2]);
        c[q * 2] = cv[0];
        c[q * 2 + 1] = cv[1];
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const F64x2 wq = *reinterpret_cast<const F64x2 *>(&sWt[h * 32 + tj * 2]);
        wv[h][0] = wq[0];
        wv[h][1] = wq[1];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        double p[4][2];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            p[h][0] = c[r] * wv[h][0];
            p[h][1] = c[r] * wv[h][1];
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            x[r][h][0] = fmax_t(x[r][h][0], p[h][0]);
            x[r][h][1] = fmax_t(x[r][h][1], p[h][1]);
        }
    }
}

// Interior tiles of a full pass, f64: see main_max_interior.
// ns: 8-pivot stages per tile (8 = one pass, 16 = a double pass); even, a runtime loop as in main_max_interior
__device__ __forceinline__ void main_max_interior_f64(double *rate, int n, int i_base, int j_base, const double *w,
                                                      const double *ct, int ct_ld, double (&sW)[2][8][128],
                                                      double (&sC)[2][8][128], int ns)
{
    constexpr int BS = 8;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int sp = tid >> 5, sv = tid & 31;
    const double *wp = w + (size_t)sp * n + j_base + sv * 4;
    const double *cp = ct + (size_t)sp * ct_ld + i_base + sv * 4;
    F64x2 pw[2], pc[2];
    auto prefetch = [&]() {
        pw[0] = *reinterpret_cast<const F64x2 *>(wp);
        pw[1] = *reinterpret_cast<const F64x2 *>(wp + 2);
        pc[0] = *reinterpret_cast<const F64x2 *>(cp);
        pc[1] = *reinterpret_cast<const F64x2 *>(cp + 2);
        wp += (size_t)BS * n;
        cp += (size_t)BS * ct_ld;
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<F64x2 *>(&sW[buf][sp][sv * 4 + 2 * u]) = pw[u];
            *reinterpret_cast<F64x2 *>(&sC[buf][sp][sv * 4 + 2 * u]) = pc[u];
        }
    };
    prefetch();
    double *xp = rate + (size_t)(i_base + ti * 8) * n + j_base + tj * 2;
    F64x2 x[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int h = 0; h < 4; ++h) x[r][h] = *reinterpret_cast<const F64x2 *>(xp + (size_t)r * n + h * 32);
    commit(0);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
#pragma unroll 1
    for (int s = 0; s < ns; s += 2) {
        prefetch();
#pragma unroll 1
        for (int t = 0; t < BS; ++t) max_fold_f64(sW[0][t], sC[0][t], ti, tj, x);
        commit(1);
        __syncthreads();
        const bool more = s + 2 < ns;                // workgroup-uniform
        if (more) prefetch();
#pragma unroll 1
        for (int t = 0; t < BS; ++t) max_fold_f64(sW[1][t], sC[1][t], ti, tj, x);
        if (more) {
            commit(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int h = 0; h < 4; ++h) *reinterpret_cast<F64x2 *>(xp + (size_t)r * n + h * 32) = x[r][h];
}

// ------------------------------------------------------------------------------------------------
// fused_main_max_f64: the rates-only main kernel at the REFERENCE'S precision (Types.hs:26,
// `_bestRate :: Double`), on the max-form domain (see fused_main_max: there the strict fold equals
// max bit for bit).  f64 has neither packed nor three-operand forms: a relaxation is one v_mul_f64
// and one v_max_f64, 8.7 measured issue cycles, and ONE wave per SIMD already saturates the f64
// pipe -- so this kernel spends its registers on the tile instead of on occupancy: 128 x 128
// entries per workgroup, 8 x 8 doubles per thread (128 VGPRs), which brings the LDS operand
// traffic down to 2 B per relaxation (the generic 8 x 4 tile reads 3 B, and at 4 waves per SIMD
// the LDS pipe was the second limit after the canonicalising v_max, see fmax_t).  8 pivots per LDS
// stage, two stages resident, the next stage prefetched into registers during the fold.
// ------------------------------------------------------------------------------------------------
template <int MINW>
__global__ __launch_bounds__(256, MINW) void fused_main_max_f64(double *rate, int rows, int n, int row0,
                                                                int k0, int bt, const double *w,
                                                                const double *ct, int ct_ld, int ct_vec,
                                                                int skip_lo, int skip_hi, ColWin cw)
{
    FWX_PROBE;
    if (cw.prio) __builtin_amdgcn_s_setprio(2);    // a launch of the look-ahead chain (FusedArgs::side)
    typedef double V2 __attribute__((ext_vector_type(2)));
    constexpr int RI = 8, NH = 4, TI = 128, TJ = 128, HJ = 32, BS = 8;
    constexpr int NST = BS / 8;                  // staging rounds per stage (8 pivots per round)
    __shared__ __attribute__((aligned(16))) double sW[2][BS][TJ];
    __shared__ __attribute__((aligned(16))) double sC[2][BS][TI];

    const int tid = threadIdx.x;
    const int i_base = blockIdx.y * TI;
    const int j_base = (blockIdx.x + cw.jt0) * TJ;
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    static_assert(NST == 1, "the interior path stages 8 pivots per round");
    {
        const int gi = row0 + i_base;
        const bool interior = (bt == B || bt == 2 * B) && ct_vec && (n & 1) == 0 && i_base + TI <= rows &&
                              j_base + TJ <= n && !(gi < j_base + TJ && j_base < gi + TI) &&
                              (k0 + bt <= j_base || k0 >= j_base + TJ) &&
                              (i_base + TI <= skip_lo || i_base >= skip_hi) && cw.clear_of(j_base, j_base + TJ);
        if (interior) {                              // workgroup-uniform
            main_max_interior_f64(rate, n, i_base, j_base, w, ct, ct_ld, sW, sC, bt / 8);
            return;
        }
    }
    const bool skip = i0 >= skip_lo && i0 < skip_hi;
    const double nanv = qnan<double>();

    // staging role: pivot sp (0..7) of the stage, 4 consecutive doubles of W and of C at sv * 4
    const int sp = tid >> 5, sv = tid & 31;
    const int sj = j_base + sv * 4, si = i_base + sv * 4;
    V2 pw[NST][2], pc[NST][2];
    auto prefetch = [&](int s0) {
#pragma unroll
        for (int g = 0; g < NST; ++g) {
            const int t = s0 + g * 8 + sp;
            const bool t_ok = t < bt;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = sj + 2 * u, i = si + 2 * u;
                pw[g][u] = (t_ok && j < n) ? *reinterpret_cast<const V2 *>(w + (size_t)t * n + j) : V2{nanv, nanv};
                if (t_ok && ct_vec && i + 2 <= rows) {
                    pc[g][u] = *reinterpret_cast<const V2 *>(ct + (size_t)t * ct_ld + i);
                } else {
                    pc[g][u][0] = (t_ok && i < rows) ? ct[(size_t)t * ct_ld + i] : nanv;
                    pc[g][u][1] = (t_ok && i + 1 < rows) ? ct[(size_t)t * ct_ld + i + 1] : nanv;
                }
                const int kcol = k0 + t - j;             // skip j == k: the pivot's own column
                if (kcol >= 0 && kcol < 2) pw[g][u][kcol] = nanv;
            }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int g = 0; g < NST; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                *reinterpret_cast<V2 *>(&sW[buf][g * 8 + sp][sv * 4 + 2 * u]) = pw[g][u];
                *reinterpret_cast<V2 *>(&sC[buf][g * 8 + sp][sv * 4 + 2 * u]) = pc[g][u];
            }
    };
    prefetch(0);

    int jcol[NH];
    bool jok[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int j = j_base + h * HJ + tj * 2;
        jok[h] = j < n && !cw.skips(j);
        jcol[h] = jok[h] ? j : n - 2;
    }
    V2 x[RI][NH];
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = min(i0 + r, rows - 1);
#pragma unroll
        for (int h = 0; h < NH; ++h)
            x[r][h] = *reinterpret_cast<const V2 *>(rate + (size_t)i * n + jcol[h]);
    }
    const int gi_lo = row0 + i_base;
    const bool diag_tile = gi_lo < j_base + TJ && j_base < gi_lo + TI;

    commit(0);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);   // the tile is complete here: see fused_main_max

    int buf = 0;
    for (int s0 = 0; s0 < bt; s0 += BS, buf ^= 1) {
        const bool more = s0 + BS < bt;
        if (more) prefetch(s0 + BS);
        const int np = skip ? 0 : min(BS, bt - s0);
        // (reading pivot t+1's operands while pivot t is folded was tried: 423 ms against 405 ms at
        // N = 16384, and 16-pivot stages 466-490 ms: profiles/r02_experiments_not_adopted.txt)
#pragma unroll 1
        for (int t = 0; t < np; ++t) max_fold_f64(sW[buf][t], sC[buf][t], ti, tj, x);
        if (more) {
            commit(buf ^ 1);
            __syncthreads();
        }
    }

    if (diag_tile) {        // the diagonal entries keep their value: restored once, with one wait
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int i = i0 + r, gi = row0 + i;
            if (i >= rows || skip) continue;
#pragma unroll
            for (int h = 0; h < NH; ++h)
                if (jok[h] && gi >= jcol[h] && gi < jcol[h] + 2) x[r][h][gi - jcol[h]] = rate[(size_t)i * n + gi];
        }
        __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
    }
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = i0 + r;
        if (i >= rows || skip) continue;
#pragma unroll
        for (int h = 0; h < NH; ++h)
            if (jok[h]) *reinterpret_cast<V2 *>(rate + (size_t)i * n + jcol[h]) = x[r][h];
    }
}

// ------------------------------------------------------------------------------------------------
// fused_main_arg: rates + next-hops (+ path trace) for f32 matrices INSIDE THE DOMAIN (fwx.h
// "Domain": every rate >= +0 and not NaN, and a non-zero rate always has a path).
//
// On that domain the strict fold equals max bit for bit (see fused_main_max), and the head of the
// reference's concatenated list (Algorithms.hs:55) is next[i][k] of the LAST successful
// relaxation.  Within one pass that relaxation is the FIRST pivot t* whose product equals the pass
// maximum -- the fold only moves on a strictly greater product, so it reaches its final value at
// the first pivot that attains it and never moves again -- and it exists iff the maximum exceeds
// the incoming value.  So instead of a compare and two selects per relaxation (12.1 issue cycles):
//   1. fold the 64 pivots two at a time with v_max3_f32, exactly as fused_main_max does (4.0), in
//      four stages of 16 pivots that ping-pong between two register tiles: comparing the tiles at
//      the end of a stage tells, per entry, the last stage `sid` in which it moved (one compare and
//      one select per 16 relaxations) -- and t* lies in that stage;
//   2. every entry that moved becomes an ITEM (entry id, sid, new value) in a per-wave list in LDS
//      (ballot + mbcnt compaction: all 64 lanes of the re-scan below do useful work);
//   3. one lane per item re-multiplies the 16 operand pairs of stage sid -- all 64 pivots of the
//      tile's C and W strips stay resident in LDS for this -- finds t* by equality, and writes
//      next = CN[t*][i] (and last = k0 + t* for the path trace).  The re-scan is INLINE at its 17
//      flush points and software-pipelined: a batch issues its gather of CN[t*][i] and the batch
//      before it stores -- as an out-of-line function every batch paid the call ABI's
//      s_waitcnt vmcnt(0) twice (gather latency at the store, store latency at the return), which
//      was most of the re-scan's cost (77 of 297 ms at N = 16384).
// About 11 % of the entries move in an average pass of the N = 16384 benchmark solve (60 % in the
// first sixteenth, 3.5 % in the last).  Bit-identical to the compare form: same products (one
// v_mul_f32 each), same winner, same t*.  No update counting (U is the number of strict increases
// along the fold: compare form only).
// ------------------------------------------------------------------------------------------------
// (8-pivot stages halve the re-scan's products and double the tracking compares: no difference,
// 257.3 vs 259.3 ms at N = 16384, gpurun_out/r02_run30.log)
constexpr int ARG_SL = 16;         // pivots per tracking stage of fused_main_arg (even, divides B)

// gfx950 needs two wait states between a vector instruction that writes a scalar register (a compare)
// and a vector instruction that reads it as a lane mask (a select), and it does not interlock.  Written
// one entry at a time -- v_cmp vcc / v_cndmask vcc -- the compiler pays an s_nop per entry: a third of
// the stage tracking, and of the pivot search of the re-scan.  These helpers do four entries per
// statement: four compares into four mask pairs, then four selects (>= 3 instructions apart).
//   track4: sid_e = (d_e == o_e) ? sid_e : stage      (the entry moved in this stage: remember it)
__device__ __forceinline__ void track4(int &s0, int &s1, int &s2, int &s3, float d0, float d1, float d2,
                                       float d3, float o0, float o1, float o2, float o3, int stage)
{
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_eq_f32 %4, %8, %12\n\tv_cmp_eq_f32 %5, %9, %13\n\t"
        "v_cmp_eq_f32 %6, %10, %14\n\tv_cmp_eq_f32 %7, %11, %15\n\t"
        "v_cndmask_b32 %0, %16, %0, %4\n\tv_cndmask_b32 %1, %16, %1, %5\n\t"
        "v_cndmask_b32 %2, %16, %2, %6\n\tv_cndmask_b32 %3, %16, %3, %7"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(stage));
}
__device__ __forceinline__ void track4(int &s0, int &s1, int &s2, int &s3, double d0, double d1, double d2,
                                       double d3, double o0, double o1, double o2, double o3, int stage)
{
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_eq_f64 %4, %8, %12\n\tv_cmp_eq_f64 %5, %9, %13\n\t"
        "v_cmp_eq_f64 %6, %10, %14\n\tv_cmp_eq_f64 %7, %11, %15\n\t"
        "v_cndmask_b32 %0, %16, %0, %4\n\tv_cndmask_b32 %1, %16, %1, %5\n\t"
        "v_cndmask_b32 %2, %16, %2, %6\n\tv_cndmask_b32 %3, %16, %3, %7"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(stage));
}
//   compact_slot: one entry slot of the compaction.  Lanes with sid >= 0 append their item id
//   (sid | id_row | E4) to the wave's list at LDS byte address lds_end (wave-uniform), in lane order;
//   returns how many did.  The compare that says "moved" writes vcc, and vcc IS the ballot, the
//   operand of both v_mbcnt and, as exec, the predicate of the store -- from C++ the ballot came back
//   through v_cndmask + v_cmp, and two branches per slot (5 vector instructions instead of 9).
template <int E4>
__device__ __forceinline__ int compact_slot(int sid, unsigned int id_row, unsigned int lds_end)
{
    int c;
    unsigned int t, id;
    unsigned long long save;
    asm volatile("v_cmp_lt_i32 vcc, -1, %[sid]\n\t"
                 "v_or3_b32 %[id], %[sid], %[row], %[e4]\n\t"
                 "s_and_saveexec_b64 %[save], vcc\n\t"          // (>= 2 instructions after the compare)
                 "v_mbcnt_lo_u32_b32 %[t], vcc_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[t], vcc_hi, %[t]\n\t"
                 "v_lshl_add_u32 %[t], %[t], 1, %[base]\n\t"
                 "ds_write_b16 %[t], %[id]\n\t"
                 "s_mov_b64 exec, %[save]\n\t"
                 "s_bcnt1_i32_b64 %[c], vcc"
                 : [c] "=s"(c), [t] "=&v"(t), [id] "=&v"(id), [save] "=&s"(save)
                 : [sid] "v"(sid), [row] "v"(id_row), [e4] "n"(E4), [base] "s"(lds_end)
                 : "vcc", "scc", "memory");
    return c;
}
//   FWX_FIND4: found = (p == m) ? U : found for four products, HIGHEST pivot first (so that, over a
//   descending sequence of calls, the smallest matching pivot wins); U0..U3 are literal pivot numbers
#define FWX_FIND4(CMP, found, m, pa, ua, pb, ub, pc_, uc, pd, ud)                                        \
    do {                                                                                                 \
        unsigned long long fm0_, fm1_, fm2_, fm3_;                                                       \
        asm(CMP " %1, %5, %9\n\t" CMP " %2, %6, %9\n\t" CMP " %3, %7, %9\n\t" CMP " %4, %8, %9\n\t"      \
            "v_cndmask_b32 %0, %0, " #ua ", %1\n\tv_cndmask_b32 %0, %0, " #ub ", %2\n\t"                \
            "v_cndmask_b32 %0, %0, " #uc ", %3\n\tv_cndmask_b32 %0, %0, " #ud ", %4"                     \
            : "+v"(found), "=&s"(fm0_), "=&s"(fm1_), "=&s"(fm2_), "=&s"(fm3_)                             \
            : "v"(pa), "v"(pb), "v"(pc_), "v"(pd), "v"(m));                                               \
    } while (0)

template <int MINW, int RI, int NP>
__global__ __launch_bounds__(256, MINW) void fused_main_arg(float *rate, int32_t *next, int rows, int n,
                                                            int row0, int k0_all, int bt_all,
                                                            const float *w_all, const float *ct_all,
                                                            const int32_t *cnt_all, int ct_ld, int ct_vec,
                                                            int skip_lo, int skip_hi, int32_t *last,
                                                            int32_t *hops, const int32_t *cht_all,
                                                            const int32_t *wh_all, ColWin cw)
{
    FWX_PROBE;
    if (cw.prio) __builtin_amdgcn_s_setprio(2);    // a launch of the look-ahead chain (FusedArgs::side)
    typedef float V4 __attribute__((ext_vector_type(4)));
    // list capacity: a flush point after every second entry slot, 63 carried + 2 * 64 new items
    constexpr int TI = 16 * RI, TJ = 64, LCAP = 192;
    // all 64 pivots of the tile's operand strips: s?[t][.], pivot t = 2 * pair + u
    __shared__ __attribute__((aligned(16))) float sW[B][TJ];
    __shared__ __attribute__((aligned(16))) float sC[B][TI];
    __shared__ unsigned short l_id[4][LCAP];    // per-wave item lists: row << 8 | column << 2 | stage
    // where the gathers of the batch in flight land (global -> LDS loads, one slot per lane):
    // CN[t*][i], and with hops CHt[t*][i] and WH[t*][j]
    __shared__ int32_t g_next[4][64], g_hc[4][64], g_hw[4][64];

    const int tid_all = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int i_base = blockIdx.y * TI;
    const int j_base = (blockIdx.x + cw.jt0) * TJ;
    if (cw.cskip_lo <= j_base && j_base + TJ <= cw.cskip_hi) return;   // the whole tile is someone else's
    const float nanv = qnan<float>();

    // the tile: loaded once (after the first pass's operand strips have been staged: see below),
    // carried in registers through every pass of the launch
    V4 xa[RI], xb[RI];

    // NP = 1 or 2 passes of up to B pivots per launch (two: the double-pass schedule of fused_range,
    // bt_all == 2 B).  A pass is complete in itself -- stage its operand strips, fold, track, re-scan
    // the moved entries, store the rows that moved -- only the tile stays in registers from one to the
    // next, which saves the second pass its tile load and the launch its second prologue, tail and gap.
    // The two passes are two inlined COPIES of the body (a lambda called twice; `#pragma unroll` refuses a
    // loop with barriers in it), not a run-time loop: around a back-edge the compiler's
    // wait-count pass sees the re-scan's global -> LDS gathers and the conditional diagonal loads as
    // pending everywhere -- s_waitcnt vmcnt(0) before every LDS store of the staging (its 12 loads
    // one at a time) and before every row store: 10 % (f32) to 20 % (f64) slower, measured
    // (tools/runs/r03_run33.sh).  Inside the body the kernel's operands are the CURRENT pass's.
    static_assert(NP == 1 || NP == 2, "passes per launch");
    auto one_pass = [&](const int ps) __attribute__((always_inline)) {
    // With two passes, everything a pass derives from the thread index is derived again in the second,
    // from a copy the compiler cannot see through: shared between the copies those values (store
    // offsets, item ids, LDS addresses -- used after the fold) would be live THROUGH the second fold,
    // 20 registers above its 148, i.e. spilled.
    int tid = tid_all;
    if (NP > 1) {
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < 256);
    }
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    const bool skip = i0 >= skip_lo && i0 < skip_hi;   // rows done by the look-ahead launch
    const int jcol = j_base + tj * 4;
    const bool jok = jcol < n && !cw.skips(jcol);
    const int jc = jok ? jcol : n - 4;
    const int k0 = k0_all + ps * B;
    const int bt = NP == 1 ? bt_all : B;
    const float *const w = w_all + (size_t)ps * B * n;
    const float *const ct = ct_all + (size_t)ps * B * ct_ld;
    const int32_t *const cnt = cnt_all + (size_t)ps * B * ct_ld;
    const int32_t *const cht = cht_all ? cht_all + (size_t)ps * B * ct_ld : nullptr;
    const int32_t *const wh = wh_all ? wh_all + (size_t)ps * B * n : nullptr;
    if (ps) __syncthreads();          // every wave has finished the re-scan that reads the previous strips

    // ---- stage W (NaN at j == k and past the matrix) and C (NaN at i == k, from colpanel) -------
    // A full pass on a tile inside the slab (all but the edge tiles): straight-line copies from two
    // pointers, no bounds, no per-vector patch -- the pivot columns, if the tile holds any, are
    // poisoned afterwards.  (The general loops below spend ~25 instructions of 64-bit address
    // arithmetic and predicates per vector, a tenth of the tile's fold.)
    const bool full_tile = bt == B && ct_vec == 1 && i_base + TI <= rows && j_base + TJ <= n;   // workgroup-uniform
    if (full_tile) {
        static_assert(TJ == 64, "16 vectors per W row");
        const float *wp = w + (size_t)(tid >> 4) * n + j_base + (tid & 15) * 4;
#pragma unroll
        for (int q = 0; q < B / 16; ++q)
            *reinterpret_cast<V4 *>(&sW[(tid >> 4) + 16 * q][(tid & 15) * 4]) =
                *reinterpret_cast<const V4 *>(wp + (size_t)(16 * q) * n);
        constexpr int CV = TI / 4, CR = 256 / CV;              // vectors per C row, rows per sweep
        const float *cp = ct + (size_t)(tid / CV) * ct_ld + i_base + (tid % CV) * 4;
#pragma unroll
        for (int q = 0; q < B / CR; ++q)
            *reinterpret_cast<V4 *>(&sC[tid / CV + CR * q][(tid % CV) * 4]) =
                *reinterpret_cast<const V4 *>(cp + (size_t)(CR * q) * ct_ld);
        if (k0 + bt > j_base && k0 < j_base + TJ) {            // skip j == k: the pivots' own columns
            __syncthreads();
            const int col = k0 + tid - j_base;
            if (tid < B && col >= 0 && col < TJ) sW[tid][col] = nanv;
        }
    } else {
    for (int idx = tid; idx < B * (TJ / 4); idx += 256) {
        const int t = idx / (TJ / 4), v = idx % (TJ / 4);
        const int j = j_base + v * 4;
        V4 val = V4{nanv, nanv, nanv, nanv};
        if (t < bt && j < n) {
            val = *reinterpret_cast<const V4 *>(w + (size_t)t * n + j);
            const int kcol = k0 + t - j;
            if (kcol >= 0 && kcol < 4) val[kcol] = nanv;
        }
        *reinterpret_cast<V4 *>(&sW[t][v * 4]) = val;
    }
    for (int idx = tid; idx < B * (TI / 4); idx += 256) {
        const int t = idx / (TI / 4), v = idx % (TI / 4);
        const int i = i_base + v * 4;
        V4 val = V4{nanv, nanv, nanv, nanv};
        if (t < bt) {
            if (ct_vec && i + 4 <= rows) {
                val = *reinterpret_cast<const V4 *>(ct + (size_t)t * ct_ld + i);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i + e < rows) val[e] = ct[(size_t)t * ct_ld + i + e];
            }
        }
        *reinterpret_cast<V4 *>(&sC[t][v * 4]) = val;
    }
    }

    if (ps == 0) {
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int i = min(i0 + r, rows - 1);
            xa[r] = *reinterpret_cast<const V4 *>(rate + (size_t)i * n + jc);
        }
    }
    int sid[RI][4];
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) sid[r][e] = -1;
    __syncthreads();

    // ---- 1. the fold: two pivots per v_max3_f32, four stages that ping-pong xa <-> xb -----------
    // wave-uniform: a wave folds unless ALL its rows belong to the look-ahead launch (its lanes
    // that do are simply never stored, see row_ok below) -- a per-lane bound would put the loop
    // counter in a vector register and the loop under an exec mask
    const int npairs = __builtin_amdgcn_ballot_w64(!skip) ? (bt + 1) / 2 : 0;
    auto pair_step = [&](int tp, const V4 (&in)[RI], V4 (&out)[RI]) {
        float c[RI][2], wv[4][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int q = 0; q < RI / 4; ++q) {
                const V4 cv = *reinterpret_cast<const V4 *>(&sC[2 * tp + u][ti * RI + q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) c[q * 4 + e][u] = cv[e];
            }
            const V4 wq = *reinterpret_cast<const V4 *>(&sW[2 * tp + u][tj * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[e][u] = wq[e];
        }
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            float p0[4], p1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm("v_mul_f32 %0, %1, %2" : "=v"(p0[e]) : "v"(c[r][0]), "v"(wv[e][0]));
                asm("v_mul_f32 %0, %1, %2" : "=v"(p1[e]) : "v"(c[r][1]), "v"(wv[e][1]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[r][e] = __builtin_fmaxf(__builtin_fmaxf(in[r][e], p0[e]), p1[e]);
        }
    };
    int stages = 0;                                  // stages executed (wave-uniform)
    constexpr int SP = ARG_SL / 2;                   // pivot pairs per stage
#pragma unroll
    for (int s = 0; s < B / ARG_SL; ++s) {
        if (s * SP < npairs) {
            const V4 (&src)[RI] = (s & 1) ? xb : xa;
            V4 (&dst)[RI] = (s & 1) ? xa : xb;
            const int p_hi = min(s * SP + SP, npairs);
            pair_step(s * SP, src, dst);
            int tp = s * SP + 1;
            if (RI == 4) {
                // the 64 x 64 form (96 VGPRs of 128): two pairs per trip, so that one wait covers the eight
                // operand reads of four pivots -- its 48-instruction pairs are too short to hide a read
#pragma unroll 1
                for (; tp + 1 < p_hi; tp += 2) {
                    pair_step(tp, dst, dst);
                    pair_step(tp + 1, dst, dst);
                }
            }
#pragma unroll 1
            for (; tp < p_hi; ++tp) pair_step(tp, dst, dst);
#pragma unroll
            for (int r = 0; r < RI; ++r)
                track4(sid[r][0], sid[r][1], sid[r][2], sid[r][3], dst[r][0], dst[r][1], dst[r][2], dst[r][3],
                       src[r][0], src[r][1], src[r][2], src[r][3], s);
            stages = s + 1;
        }
    }
    if (stages & 1) {                                // the result sits in xb: bring it to xa
#pragma unroll
        for (int r = 0; r < RI; ++r) xa[r] = xb[r];
    }

    // ---- 2. + 3. moved entries -> items -> t* -> next (and last) ---------------------------------
    const int gi_lo = row0 + i_base;
    const bool diag_tile = gi_lo < j_base + TJ && j_base < gi_lo + TI;
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    unsigned short *ids = &l_id[wave][0];
    const unsigned int ids_lds = (unsigned int)reinterpret_cast<size_t>(
        (__attribute__((address_space(3))) unsigned short *)ids);          // its LDS byte address (uniform)
    const int lane = tid & 63;
    int count = 0;                                   // items in this wave's list (wave-uniform)
    // The batch whose gathers are in flight.  They are global -> LDS loads (no destination
    // register: with one, the register allocator's copies and reuse put s_waitcnt vmcnt(0) right
    // behind the load) and the batch is retired -- LDS -> next / hops -- after the NEXT batch has
    // been scanned, or at the end.
    bool p_act = false;
    // Everything the re-scan and the row stores address, relative to the tile: a uniform base pointer
    // (scalar registers) + a 32-bit BYTE offset per lane, formed by 24-bit multiply-adds
    // (n * 4 and ct_ld * 4 < 2^24: check_fused_args) -- a 64-bit multiply-add, a move and a 64-bit
    // shift-add per access otherwise.
    char *const next_t = reinterpret_cast<char *>(next + (size_t)i_base * n + j_base);
    char *const last_t = last ? reinterpret_cast<char *>(last + (size_t)i_base * n + j_base) : nullptr;
    char *const hops_t = hops ? reinterpret_cast<char *>(hops + (size_t)i_base * n + j_base) : nullptr;
    const char *const cnt_t = reinterpret_cast<const char *>(cnt + i_base);
    const char *const cht_t = cht ? reinterpret_cast<const char *>(cht + i_base) : nullptr;
    const char *const wh_t = wh ? reinterpret_cast<const char *>(wh + j_base) : nullptr;
    char *const rate_t = reinterpret_cast<char *>(rate + (size_t)i_base * n + j_base);
    const unsigned int n4 = (unsigned int)n * 4u, ld4 = (unsigned int)ct_ld * 4u;
    unsigned int p_offb = 0;
    auto retire = [&]() __attribute__((always_inline)) {
        if (p_act) {
            // vmcnt also counts global -> LDS loads; the compiler's own wait-count pass does not put
            // this wait here (it loses track of them across the flush points' control flow)
            __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
            *reinterpret_cast<int32_t *>(next_t + p_offb) = g_next[wave][lane];
            // lengths of the two halves at the winning pivot (Algorithms.hs:55)
            if (hops)
                *reinterpret_cast<int32_t *>(hops_t + p_offb) =
                    (int32_t)((uint32_t)g_hc[wave][lane] + (uint32_t)g_hw[wave][lane]);
        }
        p_act = false;
    };
    // Re-scan of the items [0, count) in batches of 64, one lane per item; full batches only unless
    // `all`; the remainder (< 64 items) moves to the front of the list.  An item carries no value:
    // its stage is the LAST one in which the entry moved, so its final value is the maximum of that
    // stage's products, and t* the first pivot that attains it.
    // item id = row << 8 | column << 2 | stage  (column << 2 is the byte offset into a W row as it stands)
    static_assert(B / ARG_SL <= 4 && TI <= 128 && TJ <= 64 && ARG_SL == 16, "item id fields, FWX_FIND4 calls");
    auto rescan = [&](bool all) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int base = 0;
        for (; base + 64 <= count || (all && base < count); base += 64) {
            const int it = base + lane;
            const bool act = it < count;
            const unsigned int id = ids[act ? it : 0];
            const unsigned int il = id >> 8, jl4 = id & 0xFCu, t0 = (id & 3u) * ARG_SL;
            const float *pc = &sC[t0][il];
            const float *pw = reinterpret_cast<const float *>(reinterpret_cast<const char *>(&sW[t0][0]) + jl4);
            float p[ARG_SL];
#pragma unroll
            for (int u = 0; u < ARG_SL; ++u) p[u] = pc[u * TI] * pw[u * TJ];
            // (three levels of v_max3_f32 instead of a chain of eight)
            float m5[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) m5[u] = __builtin_fmaxf(__builtin_fmaxf(p[3 * u], p[3 * u + 1]), p[3 * u + 2]);
            const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m5[0], m5[1]), m5[2]),
                                            __builtin_fmaxf(__builtin_fmaxf(m5[3], m5[4]), p[15]));
            int found = -1;                          // descending: the smallest matching pivot wins
            FWX_FIND4("v_cmp_eq_f32", found, m, p[15], 15, p[14], 14, p[13], 13, p[12], 12);
            FWX_FIND4("v_cmp_eq_f32", found, m, p[11], 11, p[10], 10, p[9], 9, p[8], 8);
            FWX_FIND4("v_cmp_eq_f32", found, m, p[7], 7, p[6], 6, p[5], 5, p[4], 4);
            FWX_FIND4("v_cmp_eq_f32", found, m, p[3], 3, p[2], 2, p[1], 1, p[0], 0);
            retire();                                // the previous batch: its gathers have landed
            if (act && found >= 0) {
                const unsigned int t_abs = (unsigned int)found + t0;
                p_offb = (unsigned int)__umul24(il, n4) + jl4;
                const unsigned int c_offb = (unsigned int)__umul24(t_abs, ld4) + il * 4u;
                __builtin_amdgcn_global_load_lds((gptr_t *)(cnt_t + c_offb), (lptr_t *)&g_next[wave][0], 4, 0, 0);
                if (last) *reinterpret_cast<int32_t *>(last_t + p_offb) = k0 + (int)t_abs;
                if (hops) {
                    __builtin_amdgcn_global_load_lds((gptr_t *)(cht_t + c_offb), (lptr_t *)&g_hc[wave][0], 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t *)(wh_t + ((unsigned int)__umul24(t_abs, n4) + jl4)),
                                                     (lptr_t *)&g_hw[wave][0], 4, 0, 0);
                }
                p_act = true;
            }
        }
        const int rest = count - base;               // < 64 (<= 0 if all)
        if (rest > 0 && base > 0) {
            const unsigned short id = ids[base + (lane < rest ? lane : 0)];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) ids[lane] = id;
        }
        count = rest > 0 ? rest : 0;
    };
    // j == i is never touched (Algorithms.hs:54): a diagonal entry keeps its value and is no item.
    // Done here, once, with its own wait: a conditional load inside the loop below would make
    // every entry slot wait for ALL outstanding memory operations (vmcnt counts in order), i.e. for
    // the row stores and the pipelined gathers.
    if (diag_tile) {                                 // workgroup-uniform
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (row0 + i0 + r == jc + e) {
                    if (i0 + r < rows && sid[r][e] >= 0) xa[r][e] = rate[(size_t)(i0 + r) * n + jc + e];
                    sid[r][e] = -1;
                }
        __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
    }
    // Lanes that own nothing here -- rows past the slab, rows of the look-ahead launch, columns outside
    // the matrix or left to another launch -- drop their stage marks once (edge tiles and the
    // look-ahead rows only: wave-uniform test), so that below "moved" is just sid >= 0 and the
    // compare that says so IS the wave's ballot.
    const bool lane_ok = !skip && jok;
    if (__builtin_amdgcn_ballot_w64(!(lane_ok && i0 + RI <= rows))) {
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) sid[r][e] = (lane_ok && i0 + r < rows) ? sid[r][e] : -1;
    }
    const unsigned int lane_offb = (unsigned int)__umul24((unsigned int)(ti * RI), n4) + (unsigned int)(tj * 16);
    const unsigned int id_lane = ((unsigned int)(ti * RI) << 8) | ((unsigned int)(tj * 4) << 2);
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const unsigned int id_row = id_lane | ((unsigned int)r << 8);
        // (a moved entry => the row and the columns are inside: jc == jcol)
        if (sid[r][0] >= 0 || sid[r][1] >= 0 || sid[r][2] >= 0 || sid[r][3] >= 0)
            *reinterpret_cast<V4 *>(rate_t + (size_t)r * n4 + lane_offb) = xa[r];
        count += compact_slot<0>(sid[r][0], id_row, ids_lds + 2u * (unsigned int)count);
        count += compact_slot<4>(sid[r][1], id_row, ids_lds + 2u * (unsigned int)count);
        if (count >= 64) rescan(false);               // count <= 63 + 2 * 64 here
        count += compact_slot<8>(sid[r][2], id_row, ids_lds + 2u * (unsigned int)count);
        count += compact_slot<12>(sid[r][3], id_row, ids_lds + 2u * (unsigned int)count);
        if (count >= 64) rescan(false);
    }
    if (count > 0) rescan(true);
    retire();
    };   // one_pass
    one_pass(0);
    if constexpr (NP > 1) one_pass(1);
}

// ------------------------------------------------------------------------------------------------
// fused_main_arg_f64: the arg scheme (see fused_main_arg) at the REFERENCE'S precision -- rates +
// next-hops (+ path trace, + hops) for f64 matrices inside the domain, instead of the compare form
// (v_mul_f64 + v_cmp + three selects per relaxation).  f64 has no three-operand max: one
// v_mul_f64 + one v_max_f64 per relaxation, one pivot at a time.  All 64 pivots of the operand
// strips must stay in LDS for the re-scan, which fixes the tile: 64 x 64 (2 x 32 KB of operands +
// lists = 69 KB, two workgroups per CU), 4 x 4 entries per thread.  Its fold reads 4 B of LDS
// operands per relaxation -- at the edge of the LDS pipe -- so it is slower than
// fused_main_max_f64's, and still well ahead of the compare form.
// ------------------------------------------------------------------------------------------------
template <int MINW, int RI, int NP>
__global__ __launch_bounds__(256, MINW) void fused_main_arg_f64(double *rate, int32_t *next, int rows, int n,
                                                                int row0, int k0_all, int bt_all,
                                                                const double *w_all, const double *ct_all,
                                                                const int32_t *cnt_all, int ct_ld, int ct_vec,
                                                                int skip_lo, int skip_hi, int32_t *last,
                                                                int32_t *hops, const int32_t *cht_all,
                                                                const int32_t *wh_all, ColWin cw)
{
    FWX_PROBE;
    if (cw.prio) __builtin_amdgcn_s_setprio(2);    // a launch of the look-ahead chain (FusedArgs::side)
    typedef double V2 __attribute__((ext_vector_type(2)));
    // RI rows per thread: 64 x 64 tiles (RI = 4; 69 KB of LDS, two workgroups per CU: the shipped form)
    // or 32 x 64 (RI = 2; 52.5 KB, three: kept as an A/B switch, slower -- see launch_max_form)
    constexpr int TI = 16 * RI, TJ = 64, LCAP = 192;
    static_assert(RI == 2 || RI == 4, "tile height");
    __shared__ __attribute__((aligned(16))) double sW[B][TJ];
    __shared__ __attribute__((aligned(16))) double sC[B][TI];
    __shared__ unsigned short l_id[4][LCAP];    // per-wave item lists: row << 8 | column << 2 | stage
    __shared__ int32_t g_next[4][64], g_hc[4][64], g_hw[4][64];   // gathers of the batch in flight

    const int tid_all = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int i_base = blockIdx.y * TI;
    const int j_base = (blockIdx.x + cw.jt0) * TJ;
    if (cw.cskip_lo <= j_base && j_base + TJ <= cw.cskip_hi) return;   // the whole tile is someone else's
    const double nanv = qnan<double>();

    // the tile: loaded once, carried in registers through every pass of the launch (see fused_main_arg)
    double xa[RI][4], xb[RI][4];

    // NP = 1 or 2 passes per launch, two copies of the body; per-pass operands and (with two passes)
    // everything derived from the thread index are (re)defined inside (see fused_main_arg)
    static_assert(NP == 1 || NP == 2, "passes per launch");
    auto one_pass = [&](const int ps) __attribute__((always_inline)) {
    int tid = tid_all;
    if (NP > 1) {
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < 256);
    }
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    const bool skip = i0 >= skip_lo && i0 < skip_hi;   // rows done by the look-ahead launch
    const int k0 = k0_all + ps * B;
    const int bt = NP == 1 ? bt_all : B;
    const double *const w = w_all + (size_t)ps * B * n;
    const double *const ct = ct_all + (size_t)ps * B * ct_ld;
    const int32_t *const cnt = cnt_all + (size_t)ps * B * ct_ld;
    const int32_t *const cht = cht_all ? cht_all + (size_t)ps * B * ct_ld : nullptr;
    const int32_t *const wh = wh_all ? wh_all + (size_t)ps * B * n : nullptr;
    if (ps) __syncthreads();          // every wave has finished the re-scan that reads the previous strips

    // ---- stage W (NaN at j == k and past the matrix) and C (NaN at i == k, from colpanel) -------
    // full pass on a tile inside the slab: straight-line copies, the pivot columns poisoned afterwards
    // (see fused_main_arg)
    const bool full_tile = bt == B && ct_vec == 1 && i_base + TI <= rows && j_base + TJ <= n;   // workgroup-uniform
    if (full_tile) {
        static_assert(TJ == 64, "32 vectors per W row");
        const double *wp = w + (size_t)(tid >> 5) * n + j_base + (tid & 31) * 2;
#pragma unroll
        for (int q = 0; q < B / 8; ++q)
            *reinterpret_cast<V2 *>(&sW[(tid >> 5) + 8 * q][(tid & 31) * 2]) =
                *reinterpret_cast<const V2 *>(wp + (size_t)(8 * q) * n);
        constexpr int CV = TI / 2, CR = 256 / CV;              // vectors per C row, rows per sweep
        const double *cp = ct + (size_t)(tid / CV) * ct_ld + i_base + (tid % CV) * 2;
#pragma unroll
        for (int q = 0; q < B / CR; ++q)
            *reinterpret_cast<V2 *>(&sC[tid / CV + CR * q][(tid % CV) * 2]) =
                *reinterpret_cast<const V2 *>(cp + (size_t)(CR * q) * ct_ld);
        if (k0 + bt > j_base && k0 < j_base + TJ) {            // skip j == k: the pivots' own columns
            __syncthreads();
            const int col = k0 + tid - j_base;
            if (tid < B && col >= 0 && col < TJ) sW[tid][col] = nanv;
        }
    } else {
    for (int idx = tid; idx < B * (TJ / 2); idx += 256) {
        const int t = idx / (TJ / 2), v = idx % (TJ / 2);
        const int j = j_base + v * 2;
        V2 val = V2{nanv, nanv};
        if (t < bt && j < n) {
            val = *reinterpret_cast<const V2 *>(w + (size_t)t * n + j);
            const int kcol = k0 + t - j;
            if (kcol >= 0 && kcol < 2) val[kcol] = nanv;
        }
        *reinterpret_cast<V2 *>(&sW[t][v * 2]) = val;
    }
    for (int idx = tid; idx < B * (TI / 2); idx += 256) {
        const int t = idx / (TI / 2), v = idx % (TI / 2);
        const int i = i_base + v * 2;
        V2 val = V2{nanv, nanv};
        if (t < bt) {
            if (ct_vec && i + 2 <= rows) {
                val = *reinterpret_cast<const V2 *>(ct + (size_t)t * ct_ld + i);
            } else {
                if (i < rows) val[0] = ct[(size_t)t * ct_ld + i];
                if (i + 1 < rows) val[1] = ct[(size_t)t * ct_ld + i + 1];
            }
        }
        *reinterpret_cast<V2 *>(&sC[t][v * 2]) = val;
    }
    }

    // A thread's four columns are two 16-byte vectors HALF A TILE APART: entries e = 0, 1 at columns
    // 2 tj + e, entries e = 2, 3 at 32 + 2 tj + (e - 2).  Its two W reads per pivot then fall on 16
    // consecutive 16-byte slots across the 16 lanes of a ds_read_b128 group = all 64 banks once; with
    // four ADJACENT columns per thread the slots are 32 bytes apart and lanes tj, tj + 8 share banks
    // (2-way conflicts on every W read: 39 % of the kernel's LDS cycles, gpurun_out/r03_sq_f64).
    const int jcol = j_base + tj * 2, jcol2 = jcol + TJ / 2;
    const bool jok = jcol < n && !cw.skips(jcol);       // n % 2 == 0, window bounds % 4 == 0: whole vectors
    const bool jok2 = jcol2 < n && !cw.skips(jcol2);
    auto col_of = [&](int e) { return e < 2 ? jcol + e : jcol2 + (e - 2); };
    if (ps == 0) {                                  // the tile, once
        const int jc = jok ? jcol : n - 2, jc2 = jok2 ? jcol2 : n - 2;
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int i = min(i0 + r, rows - 1);
            const V2 a = *reinterpret_cast<const V2 *>(rate + (size_t)i * n + jc);
            const V2 b = *reinterpret_cast<const V2 *>(rate + (size_t)i * n + jc2);
            xa[r][0] = a[0]; xa[r][1] = a[1]; xa[r][2] = b[0]; xa[r][3] = b[1];
        }
    }
    int sid[RI][4];
#pragma unroll
    for (int r = 0; r < RI; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) sid[r][e] = -1;
    __syncthreads();

    // ---- 1. the fold: four stages of 16 pivots that ping-pong xa <-> xb --------------------------
    const int npiv = __builtin_amdgcn_ballot_w64(!skip) ? bt : 0;   // wave-uniform (see fused_main_arg)
    auto step = [&](int t, const double (&in)[RI][4], double (&out)[RI][4]) {
        double c[RI], wv[4];
#pragma unroll
        for (int q = 0; q < RI / 2; ++q) {
            const V2 cv = *reinterpret_cast<const V2 *>(&sC[t][ti * RI + q * 2]);
            c[q * 2] = cv[0];
            c[q * 2 + 1] = cv[1];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const V2 wq = *reinterpret_cast<const V2 *>(&sW[t][tj * 2 + q * (TJ / 2)]);
            wv[q * 2] = wq[0];
            wv[q * 2 + 1] = wq[1];
        }
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            double p[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = c[r] * wv[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) out[r][e] = fmax_t(in[r][e], p[e]);
        }
    };
    // NPV pivots from t on, in place, all their operand reads issued together (with two waves per SIMD
    // the LDS latency of every pivot is exposed; this divides the number of waits)
    auto stepn = [&](auto npv, int t, double (&x)[RI][4]) {
        constexpr int NPV = decltype(npv)::value;
        double c[NPV][RI], wv[NPV][4];
#pragma unroll
        for (int u = 0; u < NPV; ++u) {
#pragma unroll
            for (int q = 0; q < RI / 2; ++q) {
                const V2 cv = *reinterpret_cast<const V2 *>(&sC[t + u][ti * RI + q * 2]);
                c[u][q * 2] = cv[0];
                c[u][q * 2 + 1] = cv[1];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const V2 wq = *reinterpret_cast<const V2 *>(&sW[t + u][tj * 2 + q * (TJ / 2)]);
                wv[u][q * 2] = wq[0];
                wv[u][q * 2 + 1] = wq[1];
            }
        }
#pragma unroll
        for (int u = 0; u < NPV; ++u)
#pragma unroll
            for (int r = 0; r < RI; ++r) {
                double p[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = c[u][r] * wv[u][e];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[r][e] = fmax_t(x[r][e], p[e]);
            }
    };
    int stages = 0;                                  // stages executed (wave-uniform)
#pragma unroll
    for (int s = 0; s < B / ARG_SL; ++s) {
        if (s * ARG_SL < npiv) {
            const double (&src)[RI][4] = (s & 1) ? xb : xa;
            double (&dst)[RI][4] = (s & 1) ? xa : xb;
            const int t_hi = min(s * ARG_SL + ARG_SL, npiv);
            step(s * ARG_SL, src, dst);
            int t = s * ARG_SL + 1;
#pragma unroll 1
            for (; t + 1 < t_hi; t += 2) stepn(std::integral_constant<int, 2>{}, t, dst);   // (4 per trip: level, 226 VGPRs)
            if (t < t_hi) step(t, dst, dst);
#pragma unroll
            for (int r = 0; r < RI; ++r)
                track4(sid[r][0], sid[r][1], sid[r][2], sid[r][3], dst[r][0], dst[r][1], dst[r][2], dst[r][3],
                       src[r][0], src[r][1], src[r][2], src[r][3], s);
            stages = s + 1;
        }
    }
    if (stages & 1) {                                // the result sits in xb: bring it to xa
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[r][e] = xb[r][e];
    }

    // ---- 2. + 3. moved entries -> items -> t* -> next (and last, hops): as in fused_main_arg -----
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int gi_lo = row0 + i_base;
    const bool diag_tile = gi_lo < j_base + TJ && j_base < gi_lo + TI;
    unsigned short *ids = &l_id[wave][0];
    const unsigned int ids_lds = (unsigned int)reinterpret_cast<size_t>(
        (__attribute__((address_space(3))) unsigned short *)ids);          // its LDS byte address (uniform)
    const int lane = tid & 63;
    int count = 0;                                   // items in this wave's list (wave-uniform)
    bool p_act = false;
    // tile-relative addressing: uniform base pointers + 32-bit byte offsets (see fused_main_arg)
    char *const next_t = reinterpret_cast<char *>(next + (size_t)i_base * n + j_base);
    char *const last_t = last ? reinterpret_cast<char *>(last + (size_t)i_base * n + j_base) : nullptr;
    char *const hops_t = hops ? reinterpret_cast<char *>(hops + (size_t)i_base * n + j_base) : nullptr;
    const char *const cnt_t = reinterpret_cast<const char *>(cnt + i_base);
    const char *const cht_t = cht ? reinterpret_cast<const char *>(cht + i_base) : nullptr;
    const char *const wh_t = wh ? reinterpret_cast<const char *>(wh + j_base) : nullptr;
    char *const rate_t = reinterpret_cast<char *>(rate + (size_t)i_base * n + j_base);
    const unsigned int n4 = (unsigned int)n * 4u, n8 = (unsigned int)n * 8u, ld4 = (unsigned int)ct_ld * 4u;
    unsigned int p_offb = 0;
    auto retire = [&]() __attribute__((always_inline)) {
        if (p_act) {
            __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);     // the global -> LDS gathers have landed
            *reinterpret_cast<int32_t *>(next_t + p_offb) = g_next[wave][lane];
            if (hops)
                *reinterpret_cast<int32_t *>(hops_t + p_offb) =
                    (int32_t)((uint32_t)g_hc[wave][lane] + (uint32_t)g_hw[wave][lane]);
        }
        p_act = false;
    };
    // item id = row << 8 | column << 2 | stage
    static_assert(B / ARG_SL <= 4 && TI <= 128 && TJ <= 64 && ARG_SL == 16, "item id fields, FWX_FIND4 calls");
    auto rescan = [&](bool all) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int base = 0;
        for (; base + 64 <= count || (all && base < count); base += 64) {
            const int it = base + lane;
            const bool act = it < count;
            const unsigned int id = ids[act ? it : 0];
            const unsigned int il = id >> 8, jl4 = id & 0xFCu, t0 = (id & 3u) * ARG_SL;
            const double *pc = &sC[t0][il];
            const double *pw = reinterpret_cast<const double *>(reinterpret_cast<const char *>(&sW[t0][0]) + 2u * jl4);
            double p[ARG_SL];
#pragma unroll
            for (int u = 0; u < ARG_SL; ++u) p[u] = pc[u * TI] * pw[u * TJ];
            // (a tree: the 15 maxima as a chain are 15 dependent f64 instructions per batch, and with two
            //  waves per SIMD nobody else fills their latency)
            double m8[8], m4[4];
#pragma unroll
            for (int u = 0; u < 8; ++u) m8[u] = fmax_t(p[2 * u], p[2 * u + 1]);
#pragma unroll
            for (int u = 0; u < 4; ++u) m4[u] = fmax_t(m8[2 * u], m8[2 * u + 1]);
            const double m = fmax_t(fmax_t(m4[0], m4[1]), fmax_t(m4[2], m4[3]));
            int found = -1;                          // descending: the smallest matching pivot wins
            FWX_FIND4("v_cmp_eq_f64", found, m, p[15], 15, p[14], 14, p[13], 13, p[12], 12);
            FWX_FIND4("v_cmp_eq_f64", found, m, p[11], 11, p[10], 10, p[9], 9, p[8], 8);
            FWX_FIND4("v_cmp_eq_f64", found, m, p[7], 7, p[6], 6, p[5], 5, p[4], 4);
            FWX_FIND4("v_cmp_eq_f64", found, m, p[3], 3, p[2], 2, p[1], 1, p[0], 0);
            retire();                                // the previous batch
            if (act && found >= 0) {
                const unsigned int t_abs = (unsigned int)found + t0;
                p_offb = (unsigned int)__umul24(il, n4) + jl4;
                const unsigned int c_offb = (unsigned int)__umul24(t_abs, ld4) + il * 4u;
                __builtin_amdgcn_global_load_lds((gptr_t *)(cnt_t + c_offb), (lptr_t *)&g_next[wave][0], 4, 0, 0);
                if (last) *reinterpret_cast<int32_t *>(last_t + p_offb) = k0 + (int)t_abs;
                if (hops) {
                    __builtin_amdgcn_global_load_lds((gptr_t *)(cht_t + c_offb), (lptr_t *)&g_hc[wave][0], 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t *)(wh_t + ((unsigned int)__umul24(t_abs, n4) + jl4)),
                                                     (lptr_t *)&g_hw[wave][0], 4, 0, 0);
                }
                p_act = true;
            }
        }
        const int rest = count - base;               // < 64 (<= 0 if all)
        if (rest > 0 && base > 0) {
            const unsigned short id = ids[base + (lane < rest ? lane : 0)];
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) ids[lane] = id;
        }
        count = rest > 0 ? rest : 0;
    };
    // j == i is never touched (Algorithms.hs:54): restored once, with its own wait (see fused_main_arg)
    if (diag_tile) {                                 // workgroup-uniform
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((e < 2 ? jok : jok2) && row0 + i0 + r == col_of(e)) {
                    if (i0 + r < rows && sid[r][e] >= 0) xa[r][e] = rate[(size_t)(i0 + r) * n + col_of(e)];
                    sid[r][e] = -1;
                }
        __builtin_amdgcn_s_waitcnt(FWX_WAIT_VMCNT0);
    }
    // lanes that own nothing here drop their stage marks once (see fused_main_arg)
    if (__builtin_amdgcn_ballot_w64(skip || !jok || !jok2 || i0 + RI > rows)) {
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sid[r][e] = (!skip && (e < 2 ? jok : jok2) && i0 + r < rows) ? sid[r][e] : -1;
    }
    const unsigned int lane_offb = (unsigned int)__umul24((unsigned int)(ti * RI), n8) + (unsigned int)(tj * 16);
    const unsigned int id_lane = ((unsigned int)(ti * RI) << 8) | ((unsigned int)(tj * 2) << 2);
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const unsigned int id_row = id_lane | ((unsigned int)r << 8);
        const unsigned int id_row2 = id_row | ((unsigned int)(TJ / 2) << 2);     // the second vector's columns
        char *const row_p = rate_t + (size_t)r * n8 + lane_offb;
        if (sid[r][0] >= 0 || sid[r][1] >= 0) *reinterpret_cast<V2 *>(row_p) = V2{xa[r][0], xa[r][1]};
        if (sid[r][2] >= 0 || sid[r][3] >= 0)
            *reinterpret_cast<V2 *>(row_p + (TJ / 2) * 8) = V2{xa[r][2], xa[r][3]};
        count += compact_slot<0>(sid[r][0], id_row, ids_lds + 2u * (unsigned int)count);
        count += compact_slot<4>(sid[r][1], id_row, ids_lds + 2u * (unsigned int)count);
        if (count >= 64) rescan(false);               // count <= 63 + 2 * 64 here
        count += compact_slot<0>(sid[r][2], id_row2, ids_lds + 2u * (unsigned int)count);
        count += compact_slot<4>(sid[r][3], id_row2, ids_lds + 2u * (unsigned int)count);
        if (count >= 64) rescan(false);
    }
    if (count > 0) rescan(true);
    retire();
    };   // one_pass
    one_pass(0);
    if constexpr (NP > 1) one_pass(1);
}

// Domain check (fwx.h "Domain"): clears bit 0 of *flag if any rate has its sign bit set or is NaN,
// bit 1 if `next` is given and an entry with a non-zero rate has next < 0.
__global__ __launch_bounds__(256) void nonneg_check_f32(const float *rate, const int32_t *next,
                                                        size_t count, int *flag)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    bool bad = false, orphan = false;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += stride) {
        if (i + 4 <= count) {
            const uint4 v = *reinterpret_cast<const uint4 *>(rate + i);
            bad |= (v.x > 0x7F800000u) | (v.y > 0x7F800000u) | (v.z > 0x7F800000u) | (v.w > 0x7F800000u);
            if (next) {
                const int4 nv = *reinterpret_cast<const int4 *>(next + i);
                orphan |= ((v.x << 1) != 0 && nv.x < 0) | ((v.y << 1) != 0 && nv.y < 0) |
                          ((v.z << 1) != 0 && nv.z < 0) | ((v.w << 1) != 0 && nv.w < 0);
            }
        } else {
            for (size_t e = i; e < count; ++e) {
                const unsigned int b = __float_as_uint(rate[e]);
                bad |= b > 0x7F800000u;
                if (next) orphan |= (b << 1) != 0 && next[e] < 0;
            }
        }
    }
    if (bad) atomicAnd(flag, ~1);
    if (orphan) atomicAnd(flag, ~2);
}

}  // namespace

// Stage size / occupancy target per variant (LDS = BS * (TJ + TI) * sizeof(T) [+ BS*TI*4]).
template <typename T, bool HAS_NEXT> struct FusedCfg;
// The next-hop variant doubles the register tile (rate + next), so it takes a half-width tile
// (NH = 1: one 16-byte vector per thread and row) to stay at 4 waves per SIMD; with the path trace
// (a third register tile) it runs at 3 waves per SIMD rather than spill, at 2 when it also counts.
template <> struct FusedCfg<float, false> { static constexpr int BS = 32, MINW = 4, NH = 2; };
template <> struct FusedCfg<float, true> { static constexpr int BS = 16, MINW = 4, NH = 1; };
// f64: 16 pivots per stage and 3+ waves per SIMD (tools/measure_f64.py: N=16384 rates only 500 ms
// against 595 ms with 32-pivot stages at 2 waves; the compare form with next-hops 767 against 779)
template <> struct FusedCfg<double, false> { static constexpr int BS = 16, MINW = 3, NH = 2; };
template <> struct FusedCfg<double, true> { static constexpr int BS = 16, MINW = 3, NH = 1; };

// Below ~512 full-size tiles the launch is bound by the latency of ONE tile (64 pivots folded into
// 64 entries per thread); 64 x 64 tiles give 4x the workgroups and a quarter of the serial work.
static bool small_tiles(int n, int rows, long long thresh = 512)
{
    return (long long)((n + 127) / 128) * ((rows + 127) / 128) < thresh;
}
// The arg kernel keeps a tile's workgroup busy for longer (fold + re-scan) and runs 3 workgroups per
// CU: its 64 x 64 form pays off up to larger matrices (measured, rates + next: N = 4096 9.54 -> 8.8
// ms, N = 6144 23.3 -> 21.8 ms, N = 8192 unchanged; gpurun_out/r02_run15_tiles.log).
// After the re-scan rewrite (gpurun_out/r02_run60.log, 64 x 64 against 128 x 64): N = 3072 3.79 / 4.03 ms,
// 6144 18.56 / 18.88, 8192 39.17 / 38.96, 10240 70.56 / 72.02: level or ahead up to there.
static bool small_tiles_arg(int n, int rows)
{
    static const long long thresh = [] {               // FWX_ARG_SMALL_TILES_BELOW: A/B switch
        const char *e = getenv("FWX_ARG_SMALL_TILES_BELOW");
        return e ? atoll(e) : 6500LL;
    }();
    return small_tiles(n, rows, thresh);
}

__global__ __launch_bounds__(256) void nonneg_check_f64(const double *rate, const int32_t *next,
                                                        size_t count, int *flag)
{
    const size_t stride = (size_t)gridDim.x * 256;
    bool bad = false, orphan = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(rate[i]);
        bad |= b > 0x7FF0000000000000ull;
        if (next) orphan |= (b << 1) != 0 && next[i] < 0;
    }
    if (bad) atomicAnd(flag, ~1);
    if (orphan) atomicAnd(flag, ~2);
}

hipError_t launch_nonneg_check(const double *rate, const int32_t *next, size_t count, int *flag,
                               hipStream_t s)
{
    if (count == 0) return hipSuccess;
    size_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nonneg_check_f64, dim3((unsigned)blocks), dim3(256), 0, s, rate, next, count,
                       flag);
    return hipGetLastError();
}

template <> bool fused_main_starves_panels<float>(const FusedArgs<float> &a)
{
    return a.nonneg && !a.updates && a.next && small_tiles_arg(a.n, a.rows);
}
template <> bool fused_main_starves_panels<double>(const FusedArgs<double> &) { return false; }
static bool panels_32_rows_enabled()
{
    static const bool on = [] { const char *e = getenv("FWX_PANELS_32_ROWS"); return !(e && *e == '0'); }();
    static const bool tight = [] { const char *e = getenv("FWX_PANELS_TIGHT"); return !(e && *e == '0'); }();
    return on && tight;
}
template <> bool fused_panels_fit_beside<float>(const FusedArgs<float> &a)
{
    return panels_32_rows_enabled() && a.next && !a.hops && !a.plog.last && fused_main_starves_panels(a);
}
template <> bool fused_panels_fit_beside<double>(const FusedArgs<double> &) { return false; }

// f32, rates only, no update counting, domain verified by the caller: the max3 kernel.
static bool launch_max_form(const FusedArgs<float> &a, dim3 grid, dim3 block, int skip_lo,
                            int skip_hi, hipStream_t s, int32_t *last, bool small, ColWin cw, bool)
{
    if (!a.nonneg || a.updates) return false;
    int ct_vec = ((uintptr_t)a.ct % 16 == 0 && a.ct_ld % 4 == 0) ? 1 : 0;
    if (a.next) {
        // (tuning runs: FWX_ARG_GENERAL_STAGING=1 keeps the arg kernels on their general staging loops)
        static const bool general = [] { const char *e = getenv("FWX_ARG_GENERAL_STAGING"); return e && *e == '1'; }();
        if (ct_vec && general) ct_vec = 2;
        // rates + next-hops (+ trace, + hops): max-form fold, then arg re-scan of the moved entries
        // (grid.x is the caller's: all 64-column tiles, or the tiles of a column window)
        // a.bt <= B: one pass; a.bt == 2 B (check_fused_args): the two-pass instantiation
#define FWX_ARG_LAUNCH(RI_, NP_, G_)                                                                   \
        hipLaunchKernelGGL((fused_main_arg<3, RI_, NP_>), G_, block, 0, s, a.rate, a.next, a.rows, a.n,  \
                           a.row0, a.k0, a.bt, a.w, a.ct, a.cnt, a.ct_ld, ct_vec, skip_lo, skip_hi, last, \
                           a.hops, a.cht, a.wh, cw)
        if (small || small_tiles_arg(a.n, a.rows)) {
            const dim3 g(small ? grid.x : (unsigned)((a.n + 63) / 64), (unsigned)((a.rows + 63) / 64));
            if (a.bt > B) FWX_ARG_LAUNCH(4, 2, g); else FWX_ARG_LAUNCH(4, 1, g);
        } else {
            const dim3 g((unsigned)((a.n + 63) / 64), (unsigned)((a.rows + 127) / 128));
            if (a.bt > B) FWX_ARG_LAUNCH(8, 2, g); else FWX_ARG_LAUNCH(8, 1, g);
        }
#undef FWX_ARG_LAUNCH
        return true;
    }
    if (small) {
        hipLaunchKernelGGL((fused_main_max<4, 1, 4, 1>), grid, block, 0, s, a.rate, a.rows, a.n, a.row0,
                           a.k0, a.bt, a.w, a.ct, a.ct_ld, ct_vec, skip_lo, skip_hi, cw);
    } else if (small_tiles(a.n, a.rows, 3600)) {
        // mid sizes: 128 x 64 tiles.  The 768 workgroup slots of the chip quantise a launch of 128 x 128
        // tiles badly (N = 4096: 1024 tiles = 1.33 rounds), half-width tiles halve the step; their
        // extra LDS operand reads cost less than that up to N = 7168 (gpurun_out/r02_run58.log:
        // 3072: 2.65 -> 2.44 ms, 4096: 4.40 -> 4.26, 6144: 11.04 -> 10.61, 7168: 16.0 -> 15.7; from
        // 8192 on the full tile wins: 22.35 against 22.65 ms)
        const dim3 g((unsigned)((a.n + 63) / 64), (unsigned)((a.rows + 127) / 128));
        hipLaunchKernelGGL((fused_main_max<3, 1, 8, 1>), g, block, 0, s, a.rate, a.rows, a.n,
                           a.row0, a.k0, a.bt, a.w, a.ct, a.ct_ld, ct_vec, skip_lo, skip_hi, cw);
    } else {
        hipLaunchKernelGGL((fused_main_max<3, 1, 8, 2>), grid, block, 0, s, a.rate, a.rows, a.n,
                           a.row0, a.k0, a.bt, a.w, a.ct, a.ct_ld, ct_vec, skip_lo, skip_hi, cw);
    }
    return true;
}
// f64 has no packed / three-operand forms: the max form is the generic kernel with
// v_mul_f64 + v_max_f64 (2 instructions per relaxation instead of 4).
static bool launch_max_form(const FusedArgs<double> &a, dim3 grid, dim3 block, int skip_lo,
                            int skip_hi, hipStream_t s, int32_t *last, bool small, ColWin cw, bool window)
{
    if (!a.nonneg || a.updates) return false;
    if (a.next) {
        // rates + next-hops (+ trace, + hops): 64 x 64 tiles whatever the matrix order.  The caller's
        // grid counts 32-column tiles for a column window (f64 small form): two of them per tile here.
        int ct_vec = ((uintptr_t)a.ct % 16 == 0 && a.ct_ld % 2 == 0) ? 1 : 0;
        static const bool general = [] { const char *e = getenv("FWX_ARG_GENERAL_STAGING"); return e && *e == '1'; }();
        if (ct_vec && general) ct_vec = 2;
        ColWin c2 = cw;
        c2.jt0 = window ? cw.jt0 / 2 : 0;
        // 64 x 64 tiles, two workgroups per CU.  FWX_ARG_F64_SHORT_TILES=1: 32 x 64 tiles, three per CU --
        // measured SLOWER (N = 16384 + next 485.8 -> 522.0 ms, profiles/r03_experiments_not_adopted.txt
        // item 4): the third wave per SIMD buys less than the doubled W staging and LDS reads cost
        static const bool tall = [] { const char *e = getenv("FWX_ARG_F64_SHORT_TILES"); return !(e && *e == '1'); }();
        const unsigned gx = window ? grid.x / 2 : (unsigned)((a.n + 63) / 64);
#define FWX_ARG64_LAUNCH(MINW_, RI_, NP_)                                                               \
        hipLaunchKernelGGL((fused_main_arg_f64<MINW_, RI_, NP_>),                                         \
                           dim3(gx, (unsigned)((a.rows + 16 * RI_ - 1) / (16 * RI_))), block, 0, s, a.rate,  \
                           a.next, a.rows, a.n, a.row0, a.k0, a.bt, a.w, a.ct, a.cnt, a.ct_ld, ct_vec,      \
                           skip_lo, skip_hi, last, a.hops, a.cht, a.wh, c2)
        if (tall) {
            if (a.bt > B) FWX_ARG64_LAUNCH(2, 4, 2); else FWX_ARG64_LAUNCH(2, 4, 1);
        } else {
            if (a.bt > B) FWX_ARG64_LAUNCH(3, 2, 2); else FWX_ARG64_LAUNCH(3, 2, 1);
        }
#undef FWX_ARG64_LAUNCH
        return true;
    }
    if (!small) {
        const int ct_vec = ((uintptr_t)a.ct % 16 == 0 && a.ct_ld % 2 == 0) ? 1 : 0;
        const dim3 g((unsigned)((a.n + 127) / 128), (unsigned)((a.rows + 127) / 128));
        hipLaunchKernelGGL((fused_main_max_f64<2>), g, block, 0, s, a.rate, a.rows, a.n, a.row0, a.k0,
                           a.bt, a.w, a.ct, a.ct_ld, ct_vec, skip_lo, skip_hi, cw);
        return true;
    }
    hipLaunchKernelGGL((fused_main<double, false, false, 16, 2, 1, 4, true>), grid, block, 0, s,
                       a.rate, a.next, a.rows, a.n, a.row0, a.k0, a.bt, a.w, a.ct, a.cnt,
                       a.ct_ld, skip_lo, skip_hi, a.updates, nullptr, nullptr, nullptr, nullptr, cw);
    return true;
}

__global__ __launch_bounds__(256) void nonneg_check_f32_scalar(const float *rate, const int32_t *next,
                                                               size_t count, int *flag)
{
    const size_t stride = (size_t)gridDim.x * 256;
    bool bad = false, orphan = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
        const unsigned int b = __float_as_uint(rate[i]);
        bad |= b > 0x7F800000u;
        if (next) orphan |= (b << 1) != 0 && next[i] < 0;
    }
    if (bad) atomicAnd(flag, ~1);
    if (orphan) atomicAnd(flag, ~2);
}

hipError_t launch_nonneg_check(const float *rate, const int32_t *next, size_t count, int *flag,
                               hipStream_t s)
{
    if (count == 0) return hipSuccess;
    // 16-byte loads: a slab that starts at an odd row of an odd-sized matrix is read by scalars
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    if (((uintptr_t)rate % 16) || (next && ((uintptr_t)next % 16)))
        hipLaunchKernelGGL(nonneg_check_f32_scalar, dim3((unsigned)blocks), dim3(256), 0, s, rate, next,
                           count, flag);
    else
        hipLaunchKernelGGL(nonneg_check_f32, dim3((unsigned)blocks), dim3(256), 0, s, rate, next, count,
                           flag);
    return hipGetLastError();
}

template <typename T> static hipError_t check_fused_args(const FusedArgs<T> &a)
{
    constexpr int VW = Vec16<T>::W;
    // (a double pass, 2 * B pivots: the max-form and arg kernels -- inside the domain, no update count)
    const bool two = a.nonneg && !a.updates;
    if (a.next && a.bt > B && a.bt != 2 * B) return hipErrorInvalidValue;   // the arg kernels: one or two FULL passes
    if (a.bt > (two ? 2 * B : B) || a.n % VW != 0 || ((uintptr_t)a.rate % 16) || ((uintptr_t)a.w % 16) ||
        (a.next && ((uintptr_t)a.next % 16)) || a.ct_ld < a.rows)
        return hipErrorInvalidValue;
    // the arg kernels form byte offsets inside a tile with 24-bit multiplies (a row of 8-byte entries
    // must stay below 2^24 bytes; such a matrix would not fit any memory anyway)
    if (a.n >= (1 << 21) || a.ct_ld >= (1 << 21)) return hipErrorInvalidValue;
    return hipSuccess;
}

template <typename T> hipError_t launch_fused_colpanel(const FusedArgs<T> &a, hipStream_t s)
{
    if (a.rows <= 0 || a.n <= 0 || a.bt <= 0) return hipSuccess;
    if (a.bt > B) return hipErrorInvalidValue;       // a panel is one pass
    hipError_t e = check_fused_args(a);
    if (e != hipSuccess) return e;
    const dim3 cgrid((unsigned)((a.rows + 63) / 64)), block(PANEL_THREADS);
    // (the trace matrices of a slab are indexed by LOCAL row, like its rate / next)
    if (a.plog.last && !a.next) return hipErrorInvalidValue;
    if (a.hops && (!a.next || !a.wh || !a.cht)) return hipErrorInvalidValue;
#define FWX_COLPANEL(HN, HL, HH)                                                                   \
    hipLaunchKernelGGL((fused_colpanel<T, HN, HL, HH>), cgrid, block, 0, s, a.rate, a.next, a.rows, a.n, \
                       a.row0, a.k0, a.bt, a.w, a.ct, a.cnt, a.ct_ld, a.plog.last, a.plog.at_col,  \
                       a.hops, a.wh, a.cht)
    if (a.plog.last) {
        if (a.hops) FWX_COLPANEL(true, true, true); else FWX_COLPANEL(true, true, false);
    } else if (a.next) {
        if (a.hops) FWX_COLPANEL(true, false, true); else FWX_COLPANEL(true, false, false);
    } else {
        FWX_COLPANEL(false, false, false);
    }
#undef FWX_COLPANEL
    return hipGetLastError();
}

// Main kernel on local rows [r_lo, r_hi) of the slab (the colpanel must have run on them), except
// rows [skip_lo, skip_hi) (slab-local, multiples of 8; empty range = nothing skipped).
// cols.c_hi > cols.c_lo: only the columns [c_lo, c_hi) (multiples of 64; 64-column tiles);
// cols.skip_hi > cols.skip_lo: all columns but those (multiples of 4).
template <typename T>
hipError_t launch_fused_main(const FusedArgs<T> &full, int r_lo, int r_hi, hipStream_t s,
                             int skip_lo, int skip_hi, FusedCols cols)
{
    constexpr int VW = Vec16<T>::W;
    constexpr int TI = 128;
    if (r_hi <= r_lo || full.n <= 0 || full.bt <= 0) return hipSuccess;
    hipError_t e = check_fused_args(full);
    if (e != hipSuccess) return e;
    if (skip_hi > skip_lo && ((skip_lo - r_lo) % 8 != 0 || (skip_hi - r_lo) % 8 != 0))
        return hipErrorInvalidValue;
    const bool window = cols.c_hi > cols.c_lo;
    if (window && (cols.c_lo % 64 != 0 || cols.c_hi % 64 != 0 || cols.c_lo < 0 || cols.c_hi > full.n))
        return hipErrorInvalidValue;
    if (cols.skip_hi > cols.skip_lo && (cols.skip_lo % 4 != 0 || cols.skip_hi % 4 != 0))
        return hipErrorInvalidValue;
    ColWin cw;
    cw.jt0 = 0;
    cw.prio = full.side ? 1 : 0;
    cw.cskip_lo = cols.skip_hi > cols.skip_lo ? cols.skip_lo : 0;
    cw.cskip_hi = cols.skip_hi > cols.skip_lo ? cols.skip_hi : 0;
    skip_lo -= r_lo;                    // kernel sees rows relative to its own slab
    skip_hi -= r_lo;
    FusedArgs<T> a = full;
    a.rate = full.rate + (size_t)r_lo * full.n;
    a.next = full.next ? full.next + (size_t)r_lo * full.n : nullptr;
    a.rows = r_hi - r_lo;
    a.row0 = full.row0 + r_lo;
    a.ct = full.ct + r_lo;
    a.cnt = full.cnt ? full.cnt + r_lo : nullptr;
    int32_t *last = full.plog.last ? full.plog.last + (size_t)r_lo * full.n : nullptr;
    if ((last || full.hops) && !a.next) return hipErrorInvalidValue;
    a.hops = full.hops ? full.hops + (size_t)r_lo * full.n : nullptr;
    a.cht = full.cht ? full.cht + r_lo : nullptr;
    const bool track = last || a.hops;
    const dim3 block(256);
    const bool small = window || small_tiles(a.n, a.rows);      // a window is swept in 64-row tiles
    const int tj = 16 * VW * (small ? 1 : (a.next ? FusedCfg<T, true>::NH : FusedCfg<T, false>::NH));
    const int ti = small ? 64 : TI;
    dim3 grid((unsigned)((a.n + tj - 1) / tj), (unsigned)((a.rows + ti - 1) / ti));
    if (window) {                                                // tj is 64 (f32) or 32 (f64) here
        cw.jt0 = cols.c_lo / tj;
        grid.x = (unsigned)((cols.c_hi - cols.c_lo) / tj);
    }
    if (launch_max_form(a, grid, block, skip_lo, skip_hi, s, last, small, cw, window)) return hipGetLastError();
#define FWX_FUSED_LAUNCH(HN, CN, HL)                                                               \
    do {                                                                                           \
        if (small)                                                                                 \
            hipLaunchKernelGGL((fused_main<T, HN, CN, 16, 2, 1, 4, false, HL>), grid, block, 0, s, \
                               a.rate, a.next, a.rows, a.n, a.row0, a.k0, a.bt, a.w, a.ct, a.cnt,  \
                               a.ct_ld, skip_lo, skip_hi, a.updates, last, a.hops, a.cht, a.wh,    \
                               cw);                                                                \
        else                                                                                       \
            hipLaunchKernelGGL((fused_main<T, HN, CN, FusedCfg<T, HN>::BS,                         \
                                           HL ? (CN ? 2 : (FusedCfg<T, HN>::MINW > 3 ? 3 : FusedCfg<T, HN>::MINW)) \
                                              : FusedCfg<T, HN>::MINW,                             \
                                           FusedCfg<T, HN>::NH, 8, false, HL>),                    \
                               grid, block, 0, s, a.rate, a.next, a.rows, a.n, a.row0, a.k0, a.bt, \
                               a.w, a.ct, a.cnt, a.ct_ld, skip_lo, skip_hi, a.updates, last,       \
                               a.hops, a.cht, a.wh, cw);                                           \
    } while (0)
    if (track) {
        if (a.updates) FWX_FUSED_LAUNCH(true, true, true); else FWX_FUSED_LAUNCH(true, false, true);
    } else if (a.next) {
        if (a.updates) FWX_FUSED_LAUNCH(true, true, false); else FWX_FUSED_LAUNCH(true, false, false);
    } else {
        if (a.updates) FWX_FUSED_LAUNCH(false, true, false); else FWX_FUSED_LAUNCH(false, false, false);
    }
#undef FWX_FUSED_LAUNCH
    return hipGetLastError();
}

template <typename T>
hipError_t launch_fused_relax(const FusedArgs<T> &a, hipStream_t s, int skip_lo, int skip_hi)
{
    hipError_t e = launch_fused_colpanel<T>(a, s);   // skipped rows get snapshots nobody reads
    if (e != hipSuccess) return e;
    return launch_fused_main<T>(a, 0, a.rows, s, skip_lo, skip_hi, FusedCols());
}

template <typename T>
hipError_t launch_fused_panel(const T *rows_base, int n, int k0, int bt, T *w, hipStream_t s, PathLog plog,
                              const int32_t *hops_rows, int32_t *wh)
{
    if (n <= 0 || bt <= 0) return hipSuccess;
    if (bt > B || (hops_rows && !wh)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 63) / 64)), block(PANEL_THREADS);
    // plog / hops_rows point at the SAME rows as rows_base (pivot row k0 of those matrices)
#define FWX_ROWPANEL(HL, HH)                                                                       \
    hipLaunchKernelGGL((fused_rowpanel<T, HL, HH>), grid, block, 0, s, rows_base, n, k0, bt, w, plog.last, \
                       plog.at_row, hops_rows, wh)
    if (plog.last) {
        if (hops_rows) FWX_ROWPANEL(true, true); else FWX_ROWPANEL(true, false);
    } else {
        if (hops_rows) FWX_ROWPANEL(false, true); else FWX_ROWPANEL(false, false);
    }
#undef FWX_ROWPANEL
    return hipGetLastError();
}

// Row panel and column panel of pass (a.k0, a.bt) in one launch.  The slab must be the whole matrix
// (a.rows == a.n, a.row0 == 0: the column panel takes its diagonal block from the pivot rows);
// w_out / wh_out: where the snapshot panel and its hops go (a.w / a.wh are not read).
template <typename T>
hipError_t launch_fused_panels(const FusedArgs<T> &a, T *w_out, int32_t *wh_out, hipStream_t s)
{
    if (a.n <= 0 || a.bt <= 0) return hipSuccess;
    if (a.rows != a.n || a.row0 != 0 || a.bt > B || !w_out || a.ct_ld < a.rows) return hipErrorInvalidValue;
    if (a.plog.last && !a.next) return hipErrorInvalidValue;
    if (a.hops && (!a.next || !wh_out || !a.cht)) return hipErrorInvalidValue;
    const int row_wgs = (a.n + 63) / 64;
    const dim3 grid((unsigned)(2 * row_wgs)), block(PANEL_THREADS);
#define FWX_PANELS(HN, HL, HH)                                                                      \
    hipLaunchKernelGGL((fused_panels<T, HN, HL, HH>), grid, block, 0, s, row_wgs, a.rate, a.next, a.n, a.k0, \
                       a.bt, w_out, a.ct, a.cnt, a.ct_ld, a.plog.last, a.plog.at_row, a.plog.at_col,  \
                       a.hops, wh_out, a.cht)
    // f32 with next-hops, no hops: the form that fits beside two fused_main_arg workgroups (FWX_PANELS_TIGHT=0: A/B)
    static const bool tight = [] { const char *e = getenv("FWX_PANELS_TIGHT"); return !(e && *e == '0'); }();
    if constexpr (sizeof(T) == 4) {
        // (a.side:) beside a main launch whose retiring workgroups leave 36.5 KB holes (the 64 x 64 fused_main_arg): column
        // workgroups of 32 rows, 32.25 KB each (FWX_PANELS_32_ROWS=0: A/B).  With the path trace the kernel needs 36
        // registers, 4 x 40 > the 128 a retiring main workgroup frees: no point, the 64-row form stays
        if (a.side && fused_panels_fit_beside(a)) {
            const dim3 g32((unsigned)(row_wgs + (a.n + 31) / 32));
            hipLaunchKernelGGL((fused_panels_next_f32<false, 32>), g32, block, 0, s, row_wgs, a.rate, a.next, a.n, a.k0,
                               a.bt, w_out, a.ct, a.cnt, a.ct_ld, nullptr, nullptr, nullptr);
            return hipGetLastError();
        }
        if (tight && a.next && !a.hops) {
            if (a.plog.last)
                hipLaunchKernelGGL(fused_panels_next_f32<true>, grid, block, 0, s, row_wgs, a.rate, a.next, a.n, a.k0,
                                   a.bt, w_out, a.ct, a.cnt, a.ct_ld, a.plog.last, a.plog.at_row, a.plog.at_col);
            else
                hipLaunchKernelGGL(fused_panels_next_f32<false>, grid, block, 0, s, row_wgs, a.rate, a.next, a.n, a.k0,
                                   a.bt, w_out, a.ct, a.cnt, a.ct_ld, nullptr, nullptr, nullptr);
            return hipGetLastError();
        }
    }
    if (a.plog.last) {
        if (a.hops) FWX_PANELS(true, true, true); else FWX_PANELS(true, true, false);
    } else if (a.next) {
        if (a.hops) FWX_PANELS(true, false, true); else FWX_PANELS(true, false, false);
    } else if (a.nonneg) {       // rates only, inside the domain: max form
        hipLaunchKernelGGL((fused_panels<T, false, false, false, true>), grid, block, 0, s, row_wgs, a.rate,
                           a.next, a.n, a.k0, a.bt, w_out, a.ct, a.cnt, a.ct_ld, a.plog.last, a.plog.at_row,
                           a.plog.at_col, a.hops, wh_out, a.cht);
    } else {
        FWX_PANELS(false, false, false);
    }
#undef FWX_PANELS
    return hipGetLastError();
}
template hipError_t launch_fused_panels<float>(const FusedArgs<float> &, float *, int32_t *, hipStream_t);
template hipError_t launch_fused_panels<double>(const FusedArgs<double> &, double *, int32_t *, hipStream_t);

template hipError_t launch_fused_relax<float>(const FusedArgs<float> &, hipStream_t, int, int);
template hipError_t launch_fused_relax<double>(const FusedArgs<double> &, hipStream_t, int, int);
template hipError_t launch_fused_colpanel<float>(const FusedArgs<float> &, hipStream_t);
template hipError_t launch_fused_colpanel<double>(const FusedArgs<double> &, hipStream_t);
template hipError_t launch_fused_main<float>(const FusedArgs<float> &, int, int, hipStream_t, int,
                                             int, FusedCols);
template hipError_t launch_fused_main<double>(const FusedArgs<double> &, int, int, hipStream_t, int,
                                              int, FusedCols);
template hipError_t launch_fused_panel<float>(const float *, int, int, int, float *, hipStream_t, PathLog,
                                              const int32_t *, int32_t *);
template hipError_t launch_fused_panel<double>(const double *, int, int, int, double *, hipStream_t,
                                               PathLog, const int32_t *, int32_t *);

}  // namespace fwx

#ifdef FWX_CLOCK_PROBE
extern "C" __attribute__((visibility("default"))) int fwx_debug_clock(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(fwx::g_clk), 24) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[3] = {0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fwx::g_clk), z, 24) != hipSuccess) return -1;
    }
    return 0;
}
#endif

