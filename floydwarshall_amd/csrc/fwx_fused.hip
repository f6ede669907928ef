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
//   fused_diag      the B x B diagonal block through its B pivots (one workgroup, LDS):
//                   exports Wd[t][c] = D_t[k0+t][k0+c] and Cd[r][t] = D_t[k0+r][k0+t]
//   fused_rowpanel  every column j of the B pivot rows (needs Cd): exports W[t][j]
//   fused_colpanel  the B pivot columns of every row i (needs W's block columns): exports
//                   Ct[t][i] (NaN where i == k: skip i==k) and CNt[t][i] = next_t[i][k0+t]
//   fused_main      128 x 128 (f32) tile per workgroup, 8 x 8 entries per thread in registers,
//                   W tile and C tile staged in LDS: B relaxations per entry per HBM round trip
//
// Skip set: i==k via NaN in Ct, j==k via NaN injected into the staged W tile, j==i by restoring
// the diagonal entry at write-back.  Scratch copies of diagonal entries may go stale inside the
// panel kernels; they are provably never consumed (every consumer is an i==k or j==k case).
//
// Roofline: per pass 4 B read + 4 B written per entry against 3*B = 192 VALU lane-ops per entry,
// so this kernel is VALU-bound (about 3 lane-ops per relaxation), not HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fwx_kernels.h"

#pragma clang fp contract(off)

namespace fwx {

namespace {

template <typename T> __device__ __forceinline__ T qnan();
template <> __device__ __forceinline__ float qnan<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double qnan<double>() { return __builtin_nan(""); }

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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), i.e. waits for the snapshot stores each panel step issues to reach memory
// (~1 us per step, 64 steps per panel); nothing in these kernels reads those stores back.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
// The three panel kernels share one shape: 256 threads = 4 waves, each thread keeps 16 entries
// in registers (indices are compile-time constants: the t loop is fully unrolled), and the
// pivot row / pivot column of step t is exchanged through a double-buffered LDS line with ONE
// barrier per step.  All per-step operands come from LDS (latency ~100 cycles), never from L2.
// ------------------------------------------------------------------------------------------------

// Diagonal block: thread (r = tid/4, cg = tid%4) owns blk[r][16cg .. 16cg+15].
template <typename T>
__global__ __launch_bounds__(256) void fused_diag(const T *rows, int n, int k0, int bt, T *wd,
                                                  T *cdt)
{
    constexpr int Q = B / 4;
    __shared__ __attribute__((aligned(16))) T rowbuf[2][B];
    __shared__ T colbuf[2][B];
    const int tid = threadIdx.x;
    const int r = tid >> 2, cg = tid & 3;

    T d[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int c = cg * Q + q;
        d[q] = (r < bt && c < bt) ? rows[(size_t)r * n + k0 + c] : qnan<T>();
    }

#pragma unroll
    for (int t = 0; t < B; ++t) {
        if (t >= bt) continue;                    // uniform
        const int og = t / Q, oq = t % Q;         // compile-time owner of column t
        if (r == t) {
            // publish row t (time-t snapshot); its own column-t entry is the diagonal: NaN there
            // makes every candidate of column t NaN (skip j == k)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int c = cg * Q + q;
                rowbuf[t & 1][c] = (c == t) ? qnan<T>() : d[q];
                if (c != t) wd[t * B + c] = d[q];
            }
            if (cg == og) wd[t * B + t] = rows[(size_t)t * n + k0 + t];   // diagonal: never changed
        }
        if (cg == og) {
            colbuf[t & 1][r] = d[oq];             // column t at time t
            cdt[t * B + r] = d[oq];               // (entry r == t is the diagonal: never consumed)
        }
        lds_barrier();
        T cval = colbuf[t & 1][r];
        if (r == t) cval = qnan<T>();             // skip i == k
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const T cand = cval * rowbuf[t & 1][cg * Q + q];
            d[q] = (d[q] < cand) ? cand : d[q];   // the r == c entry may go stale: never consumed
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 64 columns per workgroup; wave w keeps pivot rows 16w..16w+15 of its columns in registers.
template <typename T>
__global__ __launch_bounds__(256) void fused_rowpanel(const T *rows, int n, int k0, int bt,
                                                      const T *cdt, T *w_out)
{
    constexpr int RPW = B / 4;
    __shared__ T wrow[2][64];
    __shared__ __attribute__((aligned(16))) T s_cd[B][B];     // s_cd[t][r] = D_t[k0+r][k0+t]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = blockIdx.x * 64 + lane;
    const bool valid = j < n;
    const int jc = valid ? j : n - 1;

    for (int idx = threadIdx.x; idx < B * B; idx += 256)
        s_cd[idx / B][idx % B] = (idx / B) < bt ? cdt[idx] : qnan<T>();
    __syncthreads();

    T p[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = wave * RPW + q;
        p[q] = r < bt ? rows[(size_t)r * n + jc] : qnan<T>();
    }
    // a column inside the block carries one diagonal entry, which must be published untouched
    const bool in_blk = valid && j >= k0 && j < k0 + bt;
    const T dorig = in_blk ? rows[(size_t)(j - k0) * n + j] : T(0);

#pragma unroll
    for (int t = 0; t < B; ++t) {
        if (t >= bt) continue;   // wave-uniform; `continue` keeps the loop fully unrollable so
                                 // that every p[..] index below is a compile-time constant
        const int ow = t / RPW, oq = t % RPW;
        if (wave == ow) {
            T v = p[oq];
            if (j == k0 + t) v = dorig;
            wrow[t & 1][lane] = v;
            if (valid) w_out[(size_t)t * n + j] = v;
        }
        T cd[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) cd[q] = s_cd[t][wave * RPW + q];
        lds_barrier();
        T w = wrow[t & 1][lane];
        if (j == k0 + t) w = qnan<T>();                       // skip j == k
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            if (q == oq && wave == ow) continue;              // skip i == k
            const T cand = cd[q] * w;
            p[q] = (p[q] < cand) ? cand : p[q];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 64 rows per workgroup; wave w keeps block columns 16w..16w+15 of its rows in registers.
template <typename T, bool HAS_NEXT>
__global__ __launch_bounds__(256) void fused_colpanel(const T *rate, const int32_t *next, int rows,
                                                      int n, int row0, int k0, int bt, const T *w,
                                                      T *ct, int32_t *cnt)
{
    constexpr int CPW = B / 4;
    __shared__ T ccol[2][64];
    __shared__ int32_t ncol[2][64];
    __shared__ __attribute__((aligned(16))) T s_wd[B][B];     // s_wd[t][c] = D_t[k0+t][k0+c]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int il = blockIdx.x * 64 + lane;
    const bool valid = il < rows;
    const int ic = valid ? il : rows - 1;
    const int gi = row0 + ic;

    for (int idx = threadIdx.x; idx < B * B; idx += 256) {
        const int t = idx / B, c = idx % B;
        s_wd[t][c] = (t < bt && c < bt) ? w[(size_t)t * n + k0 + c] : qnan<T>();
    }
    __syncthreads();

    T d[CPW];
    int32_t nx[CPW];
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        const int c = wave * CPW + q;
        const size_t off = (size_t)ic * n + k0 + (c < bt ? c : 0);
        d[q] = c < bt ? rate[off] : qnan<T>();
        nx[q] = (HAS_NEXT && c < bt) ? next[off] : -1;
    }

#pragma unroll
    for (int t = 0; t < B; ++t) {
        if (t >= bt) continue;   // wave-uniform (see fused_rowpanel)
        const int ow = t / CPW, oq = t % CPW;
        if (wave == ow) {
            ccol[t & 1][lane] = d[oq];
            if (HAS_NEXT) ncol[t & 1][lane] = nx[oq];
        }
        lds_barrier();
        T c = ccol[t & 1][lane];
        const int32_t cn = HAS_NEXT ? ncol[t & 1][lane] : 0;
        if (gi == k0 + t) c = qnan<T>();                      // skip i == k
        if (wave == ow && valid) {
            ct[(size_t)t * rows + il] = c;
            if (HAS_NEXT) cnt[(size_t)t * rows + il] = cn;
        }
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            if (q == oq && wave == ow) continue;              // skip j == k
            const T cand = c * s_wd[t][wave * CPW + q];
            const bool up = d[q] < cand;
            d[q] = up ? cand : d[q];
            if (HAS_NEXT) nx[q] = up ? cn : nx[q];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BS pivots are staged in LDS at a time (the register tile lives across stages), which keeps a
// workgroup at 32 KiB of LDS and <= 128 VGPRs: 4 workgroups = 16 waves per CU.  The loop is pure
// VALU (v_pk_mul_f32 + v_cmp + v_cndmask per pair of relaxations) and one wave alone issues at
// half rate on a SIMD-32, so occupancy, not bytes, is what this kernel needs.
template <typename T, bool HAS_NEXT, bool COUNT, int BS, int MINW>
__global__ __launch_bounds__(256, MINW) void fused_main(T *rate, int32_t *next, int rows, int n,
                                                        int row0, int k0, int bt, const T *w,
                                                        const T *ct, const int32_t *cnt,
                                                        unsigned long long *updates)
{
    using V = typename Vec16<T>::type;
    using IV = typename IVec<Vec16<T>::W>::type;
    constexpr int VW = Vec16<T>::W;
    constexpr int RI = 8, TI = 16 * RI;       // 128 rows
    constexpr int TJ = 16 * 2 * VW;           // 128 (f32) / 64 (f64) columns
    constexpr int HJ = TJ / 2;

    __shared__ __attribute__((aligned(16))) T sW[BS][TJ];
    __shared__ __attribute__((aligned(16))) T sC[BS][TI];
    __shared__ __attribute__((aligned(16))) int32_t sN[HAS_NEXT ? BS : 1][HAS_NEXT ? TI : 4];
    __shared__ unsigned int s_cnt;

    const int tid = threadIdx.x;
    const int i_base = blockIdx.y * TI;
    const int j_base = blockIdx.x * TJ;
    if (COUNT && tid == 0) s_cnt = 0;

    // ---- this thread's 8 x (2 vectors) register tile -------------------------------------------
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    int jcol[2];
    bool jok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j_base + h * HJ + tj * VW;
        jok[h] = j < n;
        jcol[h] = jok[h] ? j : n - VW;
    }
    V x[RI][2];
    IV nx[HAS_NEXT ? RI : 1][2];
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = min(i0 + r, rows - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            x[r][h] = *reinterpret_cast<const V *>(rate + (size_t)i * n + jcol[h]);
            if (HAS_NEXT) nx[r][h] = *reinterpret_cast<const IV *>(next + (size_t)i * n + jcol[h]);
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
            sC[tl][il] = ok ? ct[(size_t)(s0 + tl) * rows + i] : qnan<T>();
            if (HAS_NEXT) sN[tl][il] = ok ? cnt[(size_t)(s0 + tl) * rows + i] : -1;
        }
        __syncthreads();

        // ---- bs in-order relaxations per entry, operands from LDS ------------------------------
        for (int tl = 0; tl < bs; ++tl) {
            T c[RI];
            int32_t cn[HAS_NEXT ? RI : 1];
            V wv[2];
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
            wv[0] = *reinterpret_cast<const V *>(&sW[tl][tj * VW]);
            wv[1] = *reinterpret_cast<const V *>(&sW[tl][HJ + tj * VW]);
#pragma unroll
            for (int r = 0; r < RI; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < VW; ++e) {
                        const T cand = c[r] * wv[h][e];
                        const bool up = x[r][h][e] < cand;
                        if (COUNT) {
                            const bool is_diag = diag_tile && (row0 + i0 + r == jcol[h] + e);
                            my_updates += (up && !is_diag && jok[h] && i0 + r < rows) ? 1u : 0u;
                        }
                        x[r][h][e] = up ? cand : x[r][h][e];
                        if (HAS_NEXT) nx[r][h][e] = up ? cn[r] : nx[r][h][e];
                    }
        }
    }

    // ---- write back (the diagonal entry, if any, is restored from memory first) ---------------
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = i0 + r;
        if (i >= rows) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!jok[h]) continue;
            const size_t off = (size_t)i * n + jcol[h];
            if (diag_tile) {
                const int gi = row0 + i;
                if (gi >= jcol[h] && gi < jcol[h] + VW) {
                    const int e = gi - jcol[h];
                    x[r][h][e] = rate[off + e];
                    if (HAS_NEXT) nx[r][h][e] = next[off + e];
                }
            }
            *reinterpret_cast<V *>(rate + off) = x[r][h];
            if (HAS_NEXT) *reinterpret_cast<IV *>(next + off) = nx[r][h];
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
// with both products from ONE v_pk_mul_f32 (operands stored as (t, t+1) pairs in LDS):
// 1.0 VALU instruction per relaxation instead of 2.5.  The caller must have verified the domain
// (fwx_dev_check_nonneg); the next-hop variant needs the compare and stays on fused_main.
// ------------------------------------------------------------------------------------------------
template <int BS, int MINW>
__global__ __launch_bounds__(256, MINW) void fused_main_max(float *rate, int rows, int n, int row0,
                                                            int k0, int bt, const float *w,
                                                            const float *ct)
{
    typedef float V4 __attribute__((ext_vector_type(4)));
    typedef float V2 __attribute__((ext_vector_type(2)));
    constexpr int RI = 8, TI = 128, TJ = 128, HJ = 64, HP = BS / 2;

    __shared__ __attribute__((aligned(16))) V2 sW[HP][TJ];   // (W_t[j], W_{t+1}[j])
    __shared__ __attribute__((aligned(16))) V2 sC[HP][TI];   // (C_t[i], C_{t+1}[i])

    const int tid = threadIdx.x;
    const int i_base = blockIdx.y * TI;
    const int j_base = blockIdx.x * TJ;
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = i_base + ti * RI;
    int jcol[2];
    bool jok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j_base + h * HJ + tj * 4;
        jok[h] = j < n;
        jcol[h] = jok[h] ? j : n - 4;
    }
    V4 x[RI][2];
#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = min(i0 + r, rows - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            x[r][h] = *reinterpret_cast<const V4 *>(rate + (size_t)i * n + jcol[h]);
    }
    const int gi_lo = row0 + i_base, gj_lo = j_base;
    const bool diag_tile = gi_lo < gj_lo + TJ && gj_lo < gi_lo + TI;
    const float nanv = qnan<float>();

    for (int s0 = 0; s0 < bt; s0 += BS) {
        const int bs = min(BS, bt - s0);
        if (s0) __syncthreads();
        // stage (t, t+1) pairs; missing pivots and skipped operands are NaN (ignored by max)
        for (int idx = tid; idx < HP * TJ; idx += 256) {
            const int tp = idx / TJ, jl = idx % TJ;
            const int j = j_base + jl;
            V2 v;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tl = 2 * tp + u, t = s0 + tl;
                float val = nanv;
                if (tl < bs && j < n && j != k0 + t) val = w[(size_t)t * n + j];
                v[u] = val;
            }
            sW[tp][jl] = v;
        }
        for (int idx = tid; idx < HP * TI; idx += 256) {
            const int tp = idx / TI, il = idx % TI;
            const int i = i_base + il;
            V2 v;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tl = 2 * tp + u;
                v[u] = (tl < bs && i < rows) ? ct[(size_t)(s0 + tl) * rows + i] : nanv;
            }
            sC[tp][il] = v;
        }
        __syncthreads();

        const int np = (bs + 1) / 2;
        for (int tp = 0; tp < np; ++tp) {
            V2 c[RI], wv[2][4];
#pragma unroll
            for (int q = 0; q < RI / 2; ++q) {
                const V4 cv = *reinterpret_cast<const V4 *>(&sC[tp][ti * RI + q * 2]);
                c[2 * q] = V2{cv[0], cv[1]};
                c[2 * q + 1] = V2{cv[2], cv[3]};
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const V4 wq = *reinterpret_cast<const V4 *>(&sW[tp][h * HJ + tj * 4 + q * 2]);
                    wv[h][2 * q] = V2{wq[0], wq[1]};
                    wv[h][2 * q + 1] = V2{wq[2], wq[3]};
                }
#pragma unroll
            for (int r = 0; r < RI; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const V2 cand = c[r] * wv[h][e];                     // v_pk_mul_f32
                        x[r][h][e] = __builtin_fmaxf(__builtin_fmaxf(x[r][h][e], cand[0]), cand[1]);
                    }
        }
    }

#pragma unroll
    for (int r = 0; r < RI; ++r) {
        const int i = i0 + r;
        if (i >= rows) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!jok[h]) continue;
            const size_t off = (size_t)i * n + jcol[h];
            if (diag_tile) {
                const int gi = row0 + i;
                if (gi >= jcol[h] && gi < jcol[h] + 4) x[r][h][gi - jcol[h]] = rate[off + gi - jcol[h]];
            }
            *reinterpret_cast<V4 *>(rate + off) = x[r][h];
        }
    }
}

// Domain check for fused_main_max: clears *flag if any entry has its sign bit set or is NaN.
__global__ __launch_bounds__(256) void nonneg_check_f32(const float *rate, size_t count, int *flag)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    bool bad = false;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += stride) {
        if (i + 4 <= count) {
            const uint4 v = *reinterpret_cast<const uint4 *>(rate + i);
            bad |= (v.x > 0x7F800000u) | (v.y > 0x7F800000u) | (v.z > 0x7F800000u) | (v.w > 0x7F800000u);
        } else {
            for (size_t e = i; e < count; ++e) bad |= __float_as_uint(rate[e]) > 0x7F800000u;
        }
    }
    if (bad) *flag = 0;
}

}  // namespace

// Stage size / occupancy target per variant (LDS = BS * (TJ + TI) * sizeof(T) [+ BS*TI*4]).
template <typename T, bool HAS_NEXT> struct FusedCfg;
template <> struct FusedCfg<float, false> { static constexpr int BS = 32, MINW = 4; };
template <> struct FusedCfg<float, true> { static constexpr int BS = 32, MINW = 2; };
template <> struct FusedCfg<double, false> { static constexpr int BS = 32, MINW = 2; };
template <> struct FusedCfg<double, true> { static constexpr int BS = 16, MINW = 2; };

// f32, rates only, no update counting, domain verified by the caller: the max3 kernel.
static bool launch_max_form(const FusedArgs<float> &a, dim3 grid, dim3 block, hipStream_t s)
{
    if (!a.nonneg || a.next || a.updates) return false;
    hipLaunchKernelGGL((fused_main_max<32, 4>), grid, block, 0, s, a.rate, a.rows, a.n, a.row0,
                       a.k0, a.bt, a.w, a.ct);
    return true;
}
static bool launch_max_form(const FusedArgs<double> &, dim3, dim3, hipStream_t) { return false; }

hipError_t launch_nonneg_check(const float *rate, size_t count, int *flag, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(nonneg_check_f32, dim3((unsigned)blocks), dim3(256), 0, s, rate, count, flag);
    return hipGetLastError();
}

template <typename T> hipError_t launch_fused_relax(const FusedArgs<T> &a, hipStream_t s)
{
    constexpr int VW = Vec16<T>::W;
    constexpr int TI = 128, TJ = 16 * 2 * VW;
    if (a.rows <= 0 || a.n <= 0 || a.bt <= 0) return hipSuccess;
    if (a.bt > B || a.n % VW != 0 || ((uintptr_t)a.rate % 16) || ((uintptr_t)a.w % 16) ||
        (a.next && ((uintptr_t)a.next % 16)))
        return hipErrorInvalidValue;
    const dim3 cgrid((unsigned)((a.rows + 63) / 64)), block(256);
    if (a.next)
        hipLaunchKernelGGL((fused_colpanel<T, true>), cgrid, block, 0, s, a.rate, a.next, a.rows,
                           a.n, a.row0, a.k0, a.bt, a.w, a.ct, a.cnt);
    else
        hipLaunchKernelGGL((fused_colpanel<T, false>), cgrid, block, 0, s, a.rate, a.next, a.rows,
                           a.n, a.row0, a.k0, a.bt, a.w, a.ct, a.cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((a.n + TJ - 1) / TJ), (unsigned)((a.rows + TI - 1) / TI));
    if (launch_max_form(a, grid, block, s)) return hipGetLastError();
#define FWX_FUSED_LAUNCH(HN, CN)                                                                   \
    hipLaunchKernelGGL((fused_main<T, HN, CN, FusedCfg<T, HN>::BS, FusedCfg<T, HN>::MINW>), grid,  \
                       block, 0, s, a.rate, a.next, a.rows, a.n, a.row0, a.k0, a.bt, a.w, a.ct,    \
                       a.cnt, a.updates)
    if (a.next) {
        if (a.updates) FWX_FUSED_LAUNCH(true, true); else FWX_FUSED_LAUNCH(true, false);
    } else {
        if (a.updates) FWX_FUSED_LAUNCH(false, true); else FWX_FUSED_LAUNCH(false, false);
    }
#undef FWX_FUSED_LAUNCH
    return hipGetLastError();
}

template <typename T>
hipError_t launch_fused_panel(const T *rows_base, int n, int k0, int bt, T *w, T *diag_ws,
                              hipStream_t s)
{
    if (n <= 0 || bt <= 0) return hipSuccess;
    if (bt > B) return hipErrorInvalidValue;
    T *wd = diag_ws, *cdt = diag_ws + B * B;
    hipLaunchKernelGGL((fused_diag<T>), dim3(1), dim3(256), 0, s, rows_base, n, k0, bt, wd, cdt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_rowpanel<T>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s,
                       rows_base, n, k0, bt, cdt, w);
    return hipGetLastError();
}

template hipError_t launch_fused_relax<float>(const FusedArgs<float> &, hipStream_t);
template hipError_t launch_fused_relax<double>(const FusedArgs<double> &, hipStream_t);
template hipError_t launch_fused_panel<float>(const float *, int, int, int, float *, float *,
                                              hipStream_t);
template hipError_t launch_fused_panel<double>(const double *, int, int, int, double *, double *,
                                               hipStream_t);

}  // namespace fwx
