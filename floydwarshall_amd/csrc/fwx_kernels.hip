// fwx_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the max-product Floyd-Warshall
// relaxation  runAlgo  (/root/reference/src/lib/Algorithms.hs:42-61).
//
// Written for MI355X only: 64-lane wavefronts, 16-byte coalesced vector loads (1 KiB per wave
// instruction), pivot column staged in LDS, pivot row segment held in registers, rare-path
// predicated stores.  No MFMA: (max, x) with a strict compare is not a dense contraction.
//
// Kernel 1  relax_k      one launch per pivot k over a slab of rows (HBM-bound streaming read)
// Kernel 2  snapshot_row copies pivot row k into the snapshot panel (panel phase, multi-GPU)
//
// Exactness rules shared by every kernel:
//   c = r[i][k] * r[k][j]      one IEEE multiply, never contracted        (Algorithms.hs:61)
//   update iff r[i][j] < c     strict ordered compare, false on NaN       (Algorithms.hs:55)
//   skip i == k                                                            (Algorithms.hs:50)
//   skip j == i, j == k                                                    (Algorithms.hs:54)
// The i==k and j==k skips are realised by replacing the operand by NaN (NaN * x = NaN and
// `r < NaN` is false for every r), which costs nothing in the streaming loop; the j==i skip is
// checked only in the rare path that has already found `r < c`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fwx_kernels.h"

#pragma clang fp contract(off)

namespace fwx {

// Append one update record for entry `off` (= i*n+j, global) made by pivot k, in shard `shard`.
__device__ __forceinline__ void log_update(const PathLog &plog, size_t off, int k, int shard)
{
    const unsigned long long idx = plog.base[shard] + atomicAdd(&plog.count[shard], 1ull);
    if (idx < plog.base[shard + 1]) {
        plog.rec_k[idx] = k;
        plog.rec_prev[idx] = plog.head[off];
        plog.head[off] = (int32_t)idx;
    }
}

template <typename T, int W> struct VecOf;
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float, 1> { typedef float type; };
template <> struct VecOf<double, 1> { typedef double type; };

template <typename T> __device__ __forceinline__ T quiet_nan();
template <> __device__ __forceinline__ float quiet_nan<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double quiet_nan<double>() { return __builtin_nan(""); }

template <typename T, int W> struct Lanes {
    using V = typename VecOf<T, W>::type;
    static __device__ __forceinline__ T get(const V &v, int c) { return v[c]; }
    static __device__ __forceinline__ void set(V &v, int c, T x) { v[c] = x; }
    static __device__ __forceinline__ V splat(T x) { V v; for (int c = 0; c < W; ++c) v[c] = x; return v; }
};
template <typename V, bool NT, typename T> __device__ __forceinline__ V load_vec(const T *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
    return *reinterpret_cast<const V *>(p);
}

template <typename T> struct Lanes<T, 1> {
    using V = T;
    static __device__ __forceinline__ T get(const V &v, int) { return v; }
    static __device__ __forceinline__ void set(V &v, int, T x) { v = x; }
    static __device__ __forceinline__ V splat(T x) { return x; }
};

// -------------------------------------------------------------------------------------------------
// relax_k: step k of runAlgo on `rows` rows of the matrix.
//
// Work decomposition.  A workgroup (256 threads = 4 waves) owns a column strip of
// SW = 256*NV*W elements and a chunk of RPB consecutive rows.  Thread t holds, for the whole
// chunk, the NV vectors of the PIVOT ROW that cover its columns (registers), and the workgroup
// stages the chunk's PIVOT COLUMN values r[i][k] (and next[i][k], hops[i][k]) in LDS with one
// strided gather.  The streaming loop then touches HBM only for r[i][j]: each wave instruction
// reads 1 KiB of one row, UNROLL*NV such loads are in flight per thread.
//
// Stores happen only where r[i][j] < c (about 0.2 % of entries per launch at N = 16384, SURVEY.md
// Appendix B), as one exec-masked 16-byte store of the updated vector plus scalar stores of
// next/hops for the updated components.
//
// `flip` reverses the block order: launches alternate direction so that the rows streamed last
// by pivot k are streamed first by pivot k+1 and are served from the 256 MiB Infinity Cache.
// -------------------------------------------------------------------------------------------------
template <typename T, int W, int NV, int RPB, int UNROLL, bool HAS_NEXT, bool HAS_HOPS, bool COUNT,
          int MINW = 1, bool NT = false>
__global__ __launch_bounds__(256, MINW) void relax_k(T *rate, int32_t *next, int32_t *hops,
                                               const T *prow, const int32_t *phops, int rows,
                                               int n, int row0, int k, int nstrips, int flip,
                                               unsigned long long *updates, PathLog plog)
{
    using L = Lanes<T, W>;
    using V = typename L::V;
    constexpr int SW = 256 * NV * W;

    __shared__ T s_col[RPB];
    __shared__ int32_t s_ncol[HAS_NEXT ? RPB : 1];
    __shared__ int32_t s_hcol[HAS_HOPS ? RPB : 1];
    __shared__ unsigned int s_cnt;

    const int t = threadIdx.x;
    const int bid = flip ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int strip = bid % nstrips;
    const int chunk = bid / nstrips;
    const int r_begin = chunk * RPB;
    const int r_cnt = min(RPB, rows - r_begin);

    // Pivot column -> LDS (one strided gather per chunk).  Row k itself gets NaN: skip i == k.
    if (t < r_cnt) {
        const size_t off = (size_t)(r_begin + t) * n + k;
        T v = rate[off];
        if (row0 + r_begin + t == k) v = quiet_nan<T>();
        s_col[t] = v;
        if (HAS_NEXT) s_ncol[t] = next[off];
        if (HAS_HOPS) s_hcol[t] = hops[off];
    }
    if (COUNT && t == 0) s_cnt = 0;

    // Pivot row segment -> registers.  Column k gets NaN: skip j == k.  Columns past the end
    // of the row are CLAMPED to the last in-range vector and their pivot set to NaN: the
    // streaming loads stay unconditional (valid addresses) and such lanes can never update.
    V p[NV];
    int col[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c0 = strip * SW + (v * 256 + t) * W;
        if (c0 < n) {
            col[v] = c0;
            p[v] = *reinterpret_cast<const V *>(prow + c0);
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (c0 + c == k) L::set(p[v], c, quiet_nan<T>());
        } else {
            col[v] = n - W;
            p[v] = L::splat(quiet_nan<T>());
        }
    }
    __syncthreads();

    unsigned int my_updates = 0;
    T *const base = rate + (size_t)r_begin * n;

    // One row of one vector: compare, and in the rare case that something improves, store.
    auto relax_vec = [&](const V &x, const V &pv, int r, int cv) {
        const T rik = s_col[r];
        bool any = false;
        T cand[W];
#pragma unroll
        for (int c = 0; c < W; ++c) {
            cand[c] = rik * L::get(pv, c);
            any |= (L::get(x, c) < cand[c]);
        }
        if (any) {
            // Rare path: some component improves.  The diagonal (j == i) is filtered here.
            const int i = row0 + r_begin + r;
            V nx = x;
            bool changed = false;
            const size_t off = (size_t)(r_begin + r) * n + cv;
#pragma unroll
            for (int c = 0; c < W; ++c) {
                if (L::get(x, c) < cand[c] && cv + c != i) {
                    L::set(nx, c, cand[c]);
                    changed = true;
                    if (HAS_NEXT) next[off + c] = s_ncol[r];
                    if (HAS_HOPS) hops[off + c] = s_hcol[r] + phops[cv + c];
                    if (HAS_NEXT && plog.head)
                        log_update(plog, (size_t)i * n + cv + c, k, bid & (FWX_UPDATE_SHARDS_K - 1));
                    if (COUNT) ++my_updates;
                }
            }
            if (changed) *reinterpret_cast<V *>(rate + off) = nx;
        }
    };

    int r = 0;
    // Main loop: UNROLL rows x NV vectors of unconditional 16-byte loads in flight per thread.
    for (; r + UNROLL <= r_cnt; r += UNROLL) {
        V x[UNROLL][NV];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                x[u][v] = load_vec<V, NT>(base + (size_t)(r + u) * n + col[v]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                relax_vec(x[u][v], p[v], r + u, col[v]);
    }
    // Row tail (slab height not a multiple of UNROLL).
    for (; r < r_cnt; ++r) {
        V x[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v)
            x[v] = *reinterpret_cast<const V *>(base + (size_t)r * n + col[v]);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            relax_vec(x[v], p[v], r, col[v]);
    }

    if (COUNT) {
        if (my_updates) atomicAdd(&s_cnt, my_updates);
        __syncthreads();
        if (t == 0 && s_cnt)
            atomicAdd(&updates[bid & (FWX_UPDATE_SHARDS_K - 1)], (unsigned long long)s_cnt);
    }
}

// -------------------------------------------------------------------------------------------------
// small_solve: the whole of runAlgo for n <= 64 in ONE launch of one workgroup -- the reference's
// own regime (its tests stop at 4 x 4, src/test/AlgorithmsTest.hs:66-77; the README session has 4
// vertices).  rate / next / hops live in LDS for the entire solve, one barrier per pivot.  Row k
// and column k are fixed points of step k, so the in-place LDS update reads exactly the step-start
// operands (Algorithms.hs:58-60).
// -------------------------------------------------------------------------------------------------
template <typename T, bool HAS_NEXT, bool HAS_HOPS>
__global__ __launch_bounds__(256) void small_solve(T *rate, int32_t *next, int32_t *hops, int n,
                                                   int k_begin, int k_end,
                                                   unsigned long long *updates, PathLog plog)
{
    constexpr int M = FWX_SMALL_N;            // the matrix is padded to 64 x 64 with NaN
    constexpr int E = M * M / 256;            // 16 entries per thread: column c = tid % 64 fixed,
    __shared__ T R[M][M + 1];                 // rows r = tid / 64 + 4 m  (no divisions anywhere)
    __shared__ int32_t NX[HAS_NEXT ? M : 1][M + 1];
    __shared__ int32_t HP[HAS_HOPS ? M : 1][M + 1];
    __shared__ unsigned int s_cnt;
    const int tid = threadIdx.x;
    const int c = tid & 63, r0 = tid >> 6;
    if (tid == 0) s_cnt = 0;

    T x[E];
    int32_t nx[HAS_NEXT ? E : 1], hp[HAS_HOPS ? E : 1];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int r = r0 + 4 * m;
        const bool in = r < n && c < n;
        x[m] = in ? rate[(size_t)r * n + c] : quiet_nan<T>();
        R[r][c] = x[m];
        if (HAS_NEXT) { nx[m] = in ? next[(size_t)r * n + c] : -1; NX[r][c] = nx[m]; }
        if (HAS_HOPS) { hp[m] = in ? hops[(size_t)r * n + c] : 0; HP[r][c] = hp[m]; }
    }
    __syncthreads();

    unsigned int mine = 0;
    for (int k = k_begin; k < k_end; ++k) {
        // row k and column k are fixed points of step k: every operand below is a step-start value
        T rkc = R[k][c];
        const int32_t hkc = HAS_HOPS ? HP[k][c] : 0;
        if (c == k) rkc = quiet_nan<T>();                     // skip j == k
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int r = r0 + 4 * m;
            T rik = R[r][k];                                  // wave-uniform: LDS broadcast
            if (r == k || r == c) rik = quiet_nan<T>();       // skip i == k and j == i
            const T cand = rik * rkc;                         // Algorithms.hs:61
            if (x[m] < cand) {                                // :55
                x[m] = cand;
                R[r][c] = cand;
                if (HAS_NEXT) { nx[m] = NX[r][k]; NX[r][c] = nx[m]; }
                if (HAS_HOPS) { hp[m] = HP[r][k] + hkc; HP[r][c] = hp[m]; }
                if (HAS_NEXT && plog.head) log_update(plog, (size_t)r * n + c, k, 0);
                ++mine;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int r = r0 + 4 * m;
        if (r < n && c < n) {
            rate[(size_t)r * n + c] = x[m];
            if (HAS_NEXT) next[(size_t)r * n + c] = nx[m];
            if (HAS_HOPS) hops[(size_t)r * n + c] = hp[m];
        }
    }
    if (updates) {
        if (mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (tid == 0 && s_cnt) atomicAdd(&updates[0], (unsigned long long)s_cnt);
    }
}

template <typename T>
hipError_t launch_small_solve(T *rate, int32_t *next, int32_t *hops, int n, int k_begin, int k_end,
                              unsigned long long *updates, PathLog plog, hipStream_t s)
{
    if (n <= 0 || k_end <= k_begin) return hipSuccess;
    if (n > FWX_SMALL_N || (hops && !next)) return hipErrorInvalidValue;
#define FWX_SMALL(HN, HH)                                                                          \
    hipLaunchKernelGGL((small_solve<T, HN, HH>), dim3(1), dim3(256), 0, s, rate, next, hops, n,    \
                       k_begin, k_end, updates, plog)
    if (hops) FWX_SMALL(true, true);
    else if (next) FWX_SMALL(true, false);
    else FWX_SMALL(false, false);
#undef FWX_SMALL
    return hipGetLastError();
}

template hipError_t launch_small_solve<float>(float *, int32_t *, int32_t *, int, int, int,
                                              unsigned long long *, PathLog, hipStream_t);
template hipError_t launch_small_solve<double>(double *, int32_t *, int32_t *, int, int, int,
                                               unsigned long long *, PathLog, hipStream_t);

template <typename T>
__global__ __launch_bounds__(256) void snapshot_row(T *dst, const T *src, int32_t *hdst,
                                                    const int32_t *hsrc, int n)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) {
        dst[j] = src[j];
        if (hdst) hdst[j] = hsrc[j];
    }
}

// -------------------------------------------------------------------------------------------------
// Host-side launchers
// -------------------------------------------------------------------------------------------------
template <typename T, int W, int NV, int RPB, int UNROLL, int MINW = 1, bool NT = false>
static hipError_t launch_relax_cfg(const RelaxArgs<T> &a, hipStream_t s)
{
    constexpr int SW = 256 * NV * W;
    const int nstrips = (a.n + SW - 1) / SW;
    const int nchunks = (a.rows + RPB - 1) / RPB;
    const dim3 grid((unsigned)(nstrips * nchunks)), block(256);
    if (grid.x == 0) return hipSuccess;
#define FWX_LAUNCH(HN, HH, CN)                                                                     \
    hipLaunchKernelGGL((relax_k<T, W, NV, RPB, UNROLL, HN, HH, CN, MINW, NT>), grid, block, 0, s, a.rate,    \
                       a.next, a.hops, a.prow, a.phops, a.rows, a.n, a.row0, a.k, nstrips,         \
                       a.flip, a.updates, a.plog)
    const bool hn = a.next != nullptr, hh = a.hops != nullptr, cn = a.updates != nullptr;
    if (hh) {
        if (cn) FWX_LAUNCH(true, true, true); else FWX_LAUNCH(true, true, false);
    } else if (hn) {
        if (cn) FWX_LAUNCH(true, false, true); else FWX_LAUNCH(true, false, false);
    } else {
        if (cn) FWX_LAUNCH(false, false, true); else FWX_LAUNCH(false, false, false);
    }
#undef FWX_LAUNCH
    return hipGetLastError();
}

template <typename T> hipError_t launch_relax(const RelaxArgs<T> &a, hipStream_t s)
{
    constexpr int WV = 16 / (int)sizeof(T);
    const bool vec_ok = (a.n % WV == 0) && ((uintptr_t)a.rate % 16 == 0) &&
                        ((uintptr_t)a.prow % 16 == 0);
    if (a.hops && !a.next) return hipErrorInvalidValue;  // hops ride on the next-hop path
    // Launch geometry from the sweep in tools/tune_relax.hip (profiles/r01_tune_relax.txt):
    // one 16-byte vector per thread (strip = 1024 f32 / 512 f64 columns), 4 rows per workgroup,
    // 4 loads in flight per thread.  Many small workgroups beat fewer large ones by 10-15 % at
    // N = 16384: the resident set then covers a compact band of rows (DRAM page locality) and
    // the tail of the launch is short.
    if (!vec_ok) return launch_relax_cfg<T, 1, 1, 4, 4>(a, s);
    return launch_relax_cfg<T, WV, 1, 4, 4>(a, s);
}

template hipError_t launch_relax<float>(const RelaxArgs<float> &, hipStream_t);
template hipError_t launch_relax<double>(const RelaxArgs<double> &, hipStream_t);

template <typename T>
hipError_t launch_snapshot_row(T *dst, const T *src, int32_t *hdst, const int32_t *hsrc, int n,
                               hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((snapshot_row<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst,
                       src, hdst, hsrc, n);
    return hipGetLastError();
}

template hipError_t launch_snapshot_row<float>(float *, const float *, int32_t *, const int32_t *,
                                               int, hipStream_t);
template hipError_t launch_snapshot_row<double>(double *, const double *, int32_t *,
                                                const int32_t *, int, hipStream_t);

}  // namespace fwx
