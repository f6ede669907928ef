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
                                               const T *prow, const int32_t *phops,
                                               const int32_t *pnext, int rows,
                                               int n, int row0, int k, int nstrips, int flip,
                                               unsigned long long *updates, PathLog plog,
                                               int skip_lo, int skip_hi)
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
    if (r_begin >= skip_lo && r_begin < skip_hi) return;   // whole workgroup: rows already relaxed

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
    if (HAS_NEXT && plog.last) {
        // snapshots of `last` for step k (see PathLog): column k by the first strip's workgroups,
        // row k by the first chunk's -- neither is modified during this launch
        if (strip == 0 && t < r_cnt) {
            const size_t off = (size_t)(row0 + r_begin + t) * n + k;
            plog.at_col[off] = plog.last[off];
        }
        if (chunk == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c0 = strip * SW + (v * 256 + t) * W;
#pragma unroll
                for (int c = 0; c < W; ++c)
                    if (c0 + c < n) plog.at_row[(size_t)k * n + c0 + c] = plog.last[(size_t)k * n + c0 + c];
            }
        }
    }

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
                    if (HAS_NEXT) {     // head (ikPath ++ kjPath), Algorithms.hs:55
                        const int32_t nik = s_ncol[r];
                        next[off + c] = (nik >= 0 || !pnext) ? nik : pnext[cv + c];
                    }
                    if (HAS_HOPS) hops[off + c] = s_hcol[r] + phops[cv + c];
                    if (HAS_NEXT && plog.last) plog.last[(size_t)i * n + cv + c] = k;
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
// small_solve: the whole of runAlgo for n <= 128 in ONE launch of one workgroup -- the reference's
// own regime (its tests stop at 4 x 4, src/test/AlgorithmsTest.hs:66-77; the README session has 4
// vertices; a market of 10 exchanges x 12 currencies has 120).  The matrix is padded with NaN to
// M x M (M = 64 or 128) and lives in REGISTERS for the entire solve (1024 threads; the 128-wide
// tile keeps its index matrices in LDS): thread (r0, c) holds column c of the rows r0, r0+RG, ...  Only pivot row k and pivot column k pass through LDS, double
// buffered: during step k the threads that own entries of row k+1 / column k+1 publish their
// post-step values (= the time-(k+1) operands) into the other buffer, so one barrier per pivot is
// enough.  Row k and column k are fixed points of step k, so what step k reads are exactly the
// step-start operands (Algorithms.hs:58-60).
// -------------------------------------------------------------------------------------------------
template <typename T, int M, int RG, bool HAS_NEXT, bool HAS_HOPS, bool LOG>
__global__ __launch_bounds__(M * RG) void small_solve(T *rate, int32_t *next, int32_t *hops, int n,
                                                      int k_begin, int k_end,
                                                      unsigned long long *updates, PathLog plog)
{
    constexpr int E = M / RG;                 // entries per thread; rows r = r0 + RG*m
    constexpr int G = 4;                      // entries per branch-free group
    constexpr int LOG_M = M == 64 ? 6 : 7;
    // Where next / hops live.  64-wide tile: in registers like the rates (4 entries per thread).
    // 128-wide tile: 16 entries per thread and 128 registers each, so the two index matrices
    // stay in LDS (2 x 66 KB of the CU's 160 KB) and only the rates are in registers; row k and
    // column k of an LDS-resident matrix are read in place (they are not written during step k).
    constexpr bool IDXL = M == 128;
    constexpr bool NXL = IDXL && HAS_NEXT, HPL = IDXL && HAS_HOPS;
    static_assert(M == 64 || M == 128, "M");
    static_assert(E % G == 0, "E");
    __shared__ T rowR[2][M], colR[2][M];      // pivot row k / pivot column k at time k
    __shared__ int32_t rowH[2][HAS_HOPS && !IDXL ? M : 1], colH[2][HAS_HOPS && !IDXL ? M : 1];
    __shared__ int32_t colN[2][HAS_NEXT && !IDXL ? M : 1], rowN[2][HAS_NEXT && !IDXL ? M : 1];
    __shared__ int32_t NX[NXL ? M : 1][NXL ? M + 1 : 1], HP[HPL ? M : 1][HPL ? M + 1 : 1];
    __shared__ unsigned int s_cnt;
    const int tid = threadIdx.x;
    // a wave never straddles two rows: r0 is wave-uniform, say so (scalar row tests, fewer VGPRs)
    const int c = tid & (M - 1), r0 = __builtin_amdgcn_readfirstlane(tid >> LOG_M);
    if (tid == 0) s_cnt = 0;
    const int off0 = r0 * n + c, off_step = RG * n;   // entry (r0 + RG*m, c) is at off0 + m*off_step
    // LOG: the path trace (see PathLog).  hd[m] = pivot of the newest update of this thread's m-th
    // entry; its snapshots for step k are stored by the owners of row k / column k when they
    // publish their operands, the final values at the end.

    T x[E];
    int32_t nx[HAS_NEXT && !IDXL ? E : 1], hp[HAS_HOPS && !IDXL ? E : 1], hd[LOG ? E : 1];
    auto publish = [&](int k, int b) {        // my entries of row k / column k -> buffer b
#pragma unroll
        for (int m = 0; m < E; ++m) {
            if (r0 + RG * m == k) {           // scalar
                rowR[b][c] = x[m];
                if constexpr (HAS_NEXT && !IDXL) rowN[b][c] = nx[m];
                if constexpr (HAS_HOPS && !IDXL) rowH[b][c] = hp[m];
                if constexpr (LOG) { if (c < n) plog.at_row[(size_t)k * n + c] = hd[m]; }
            }
        }
        if (c == k) {                         // one lane of the wave that holds column k
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const int r = r0 + RG * m;
                colR[b][r] = r == k ? quiet_nan<T>() : x[m];   // skip i == k: NaN at the source
                if constexpr (HAS_NEXT && !IDXL) colN[b][r] = nx[m];
                if constexpr (HAS_HOPS && !IDXL) colH[b][r] = hp[m];
                if constexpr (LOG) { if (r < n) plog.at_col[(size_t)r * n + k] = hd[m]; }
            }
        }
    };
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int r = r0 + RG * m;
        const bool in = r < n && c < n;
        x[m] = in ? rate[off0 + m * off_step] : quiet_nan<T>();
        // skip j == i: a diagonal entry is never an operand (it could only be one in the steps
        // that skip it) and never a target, so its register holds +inf -- no candidate compares
        // greater -- and the value in memory is left as it is
        if (r == c) x[m] = (T)__builtin_huge_val();
        if constexpr (HAS_NEXT) {
            const int32_t v = in ? next[off0 + m * off_step] : -1;
            if constexpr (IDXL) NX[r][c] = v; else nx[m] = v;
        }
        if constexpr (HAS_HOPS) {
            const int32_t v = in ? hops[off0 + m * off_step] : 0;
            if constexpr (IDXL) HP[r][c] = v; else hp[m] = v;
        }
        if constexpr (LOG) hd[m] = -1;
    }
    publish(k_begin, k_begin & 1);
    __syncthreads();

    unsigned int mine = 0;
    for (int k = k_begin; k < k_end; ++k) {
        const int b = k & 1;
        T rkc = rowR[b][c];
        int32_t hkc = 0, nkc = -1;
        if constexpr (HPL) hkc = HP[k][c];
        else if constexpr (HAS_HOPS) hkc = rowH[b][c];
        // head kjPath, for the (off-domain) case of an update whose ikPath is empty
        if constexpr (NXL) nkc = NX[k][c];
        else if constexpr (HAS_NEXT) nkc = rowN[b][c];
        if (c == k) rkc = quiet_nan<T>();                     // skip j == k
#pragma unroll
        for (int g = 0; g < E; g += G) {
            if (r0 + RG * g >= n) break;                      // scalar: nothing but padding rows left
            // G entries: the pivot-column operands are read unconditionally (so the LDS reads of
            // the group overlap) and the update is a select / predicated store, exactly
            //   if (x < cand) { x = cand; next = next[i][k]; hops = hops[i][k] + hops[k][j]; }
#pragma unroll
            for (int m = g; m < g + G; ++m) {
                const int r = r0 + RG * m;
                const T raw = colR[b][r];                     // wave-uniform address: LDS broadcast
                int32_t cn = 0, ch = 0;
                if constexpr (NXL) cn = NX[r][k]; else if constexpr (HAS_NEXT) cn = colN[b][r];
                if constexpr (HAS_NEXT) {
                    // head (ikPath ++ kjPath): next[i][k] unless ikPath is empty (Algorithms.hs:55).
                    // r is wave-uniform, so this is a scalar test and a rarely taken move.
                    if (__builtin_amdgcn_readfirstlane(cn) < 0) cn = nkc;
                }
                if constexpr (HPL) ch = HP[r][k]; else if constexpr (HAS_HOPS) ch = colH[b][r];
                const T cand = raw * rkc;                     // Algorithms.hs:61
                const bool p = x[m] < cand;                   // :55 (false on NaN)
                x[m] = p ? cand : x[m];
                if constexpr (!IDXL) {
                    if constexpr (HAS_NEXT) nx[m] = p ? cn : nx[m];
                    if constexpr (HAS_HOPS) hp[m] = p ? ch + hkc : hp[m];
                } else if (p) {
                    if constexpr (HAS_NEXT) NX[r][c] = cn;
                    if constexpr (HAS_HOPS) HP[r][c] = ch + hkc;
                }
                mine += (unsigned int)__builtin_popcountll(__ballot(p));   // scalar; wave total
                if constexpr (LOG) hd[m] = p ? k : hd[m];
            }
        }
        if (k + 1 < k_end) publish(k + 1, b ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int r = r0 + RG * m;
        if (r < n && c < n) {
            if (r != c) rate[off0 + m * off_step] = x[m];
            if constexpr (NXL) next[off0 + m * off_step] = NX[r][c];
            else if constexpr (HAS_NEXT) next[off0 + m * off_step] = nx[m];
            if constexpr (HPL) hops[off0 + m * off_step] = HP[r][c];
            else if constexpr (HAS_HOPS) hops[off0 + m * off_step] = hp[m];
            if constexpr (LOG) plog.last[off0 + m * off_step] = hd[m];
        }
    }
    if (updates) {
        if (mine && (tid & 63) == 0) atomicAdd(&s_cnt, mine);   // `mine` is a wave total
        __syncthreads();
        if (tid == 0 && s_cnt) atomicAdd(&updates[0], (unsigned long long)s_cnt);
    }
}

template <typename T>
hipError_t launch_small_solve(T *rate, int32_t *next, int32_t *hops, int n, int k_begin, int k_end,
                              unsigned long long *updates, PathLog plog, hipStream_t s)
{
    if (n <= 0 || k_end <= k_begin) return hipSuccess;
    if (n > FWX_SMALL_N || (hops && !next) || k_begin < 0 || k_end > n) return hipErrorInvalidValue;
#define FWX_SMALL(M, RG, HN, HH, LG)                                                               \
    hipLaunchKernelGGL((small_solve<T, M, RG, HN, HH, LG>), dim3(1), dim3(M * RG), 0, s, rate,     \
                       next, hops, n, k_begin, k_end, updates, plog)
#define FWX_SMALL_M(M, RG)                                                                         \
    do {                                                                                           \
        const bool lg = next && plog.last;                                                         \
        if (hops) { if (lg) FWX_SMALL(M, RG, true, true, true); else FWX_SMALL(M, RG, true, true, false); }    \
        else if (next) { if (lg) FWX_SMALL(M, RG, true, false, true); else FWX_SMALL(M, RG, true, false, false); } \
        else FWX_SMALL(M, RG, false, false, false);                                                \
    } while (0)
    if (n <= 64) FWX_SMALL_M(64, 16);         // 1024 threads x 4 entries: 4 waves per SIMD hide
    else FWX_SMALL_M(128, 8);                 //   the LDS latency; 1024 threads x 16 entries
#undef FWX_SMALL_M
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
    if (a.skip_hi > a.skip_lo && (a.skip_lo % RPB || a.skip_hi % RPB)) return hipErrorInvalidValue;
#define FWX_LAUNCH(HN, HH, CN)                                                                     \
    hipLaunchKernelGGL((relax_k<T, W, NV, RPB, UNROLL, HN, HH, CN, MINW, NT>), grid, block, 0, s, a.rate,    \
                       a.next, a.hops, a.prow, a.phops, a.pnext, a.rows, a.n, a.row0, a.k, nstrips, \
                       a.flip, a.updates, a.plog, a.skip_lo, a.skip_hi)
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
    // Launch geometry from the sweep in tools/tune_relax.hip (profiles/r01_tune_relax_sweep1.txt ... _sweep3.txt):
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
