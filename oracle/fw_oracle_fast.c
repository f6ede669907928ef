/*
 * fw_oracle_fast.c -- the oracle's multi-threaded loop (fw_oracle.c, fwo_relax_mt_*) with a chunk pre-check,
 * for the tests that continue a solve at N = 8192 ... 32768 (hundreds of pivots over gigabytes).
 *
 * TEST INFRASTRUCTURE, like fw_oracle.c: only tests/ load it; the product never does, and the CPU baseline
 * of bench.py stays the plain loop of fw_oracle.c.
 *
 * A chunk of 64 columns of row i is SKIPPED in step k iff no j in it satisfies ri[j] < rik * rk[j] -- the
 * very compare of /root/reference/src/lib/Algorithms.hs:55, evaluated without side effects (a loop the
 * compiler vectorises: this file is built at -O3, and its worker also for AVX2, dispatched at load time --
 * the library is built in one container and run on another host; no floating-point reduction, nothing
 * re-associated or fused: -ffp-contract=off, no fast-math).  A chunk with at least one such j (the skip
 * columns j == i, j == k may raise a false alarm, never hide a hit) runs the loop of fw_oracle.c unchanged:
 * Algorithms.hs:50 (row k copied), :54 (j == i, j == k untouched), :55 (strict compare; path = ikPath ++
 * kjPath: head = next[i][k], or next[k][j] when ikPath is empty; length = len ik + len kj), :58-61 (operands
 * from the start of step k: row k and column k are fixed points of the step).  Skipping a chunk in which
 * nothing would have been written is the identity: same results, same U.  tests/test_oracle_golden.py pins
 * it to fwo_relax_* / fwo_relax_mt_* on ordinary, tie-heavy, sparse and hostile (inf / NaN / negative) inputs.
 */
#include <pthread.h>
#include <stdint.h>

typedef struct {
    int32_t n, k_begin, k_end, tid, threads;
    void *rate;
    int32_t *next;
    int32_t *hops;
    pthread_barrier_t *bar;
    uint64_t updates;
} fwo_fast_job;

#define FWO_CHUNK 64
#define FWO_DEFINE_WORKER_FAST(NAME, T)                                                            \
    __attribute__((target_clones("avx2", "default"))) static void *NAME(void *arg)                 \
    {                                                                                              \
        fwo_fast_job *job = (fwo_fast_job *)arg;                                                   \
        const int32_t n = job->n;                                                                  \
        const size_t N = (size_t)n;                                                                \
        T *rate = (T *)job->rate;                                                                  \
        int32_t *next = job->next;                                                                 \
        int32_t *hops = job->hops;                                                                 \
        const int32_t lo = (int32_t)(((int64_t)n * job->tid) / job->threads);                      \
        const int32_t hi = (int32_t)(((int64_t)n * (job->tid + 1)) / job->threads);                \
        uint64_t updates = 0;                                                                      \
        for (int32_t k = job->k_begin; k < job->k_end; ++k) {                                      \
            const T *rk = rate + (size_t)k * N;                                                    \
            for (int32_t i = lo; i < hi; ++i) {                                                    \
                if (i == k) continue;                                                              \
                T *ri = rate + (size_t)i * N;                                                      \
                const T rik = ri[k];                                                               \
                const int32_t nik = next ? next[(size_t)i * N + k] : 0;                            \
                const int32_t hik = hops ? hops[(size_t)i * N + k] : 0;                            \
                for (int32_t j0 = 0; j0 < n; j0 += FWO_CHUNK) {                                    \
                    const int32_t j1 = j0 + FWO_CHUNK < n ? j0 + FWO_CHUNK : n;                    \
                    int any = 0;                                                                   \
                    for (int32_t j = j0; j < j1; ++j) any |= ri[j] < rik * rk[j];                  \
                    if (!any) continue;                                                            \
                    for (int32_t j = j0; j < j1; ++j) {                                            \
                        if (j == i || j == k) continue;                                            \
                        const T c = rik * rk[j];                                                   \
                        if (ri[j] < c) {                                                           \
                            ri[j] = c;                                                             \
                            if (next)                                                              \
                                next[(size_t)i * N + j] = nik >= 0 ? nik : next[(size_t)k * N + j];\
                            if (hops) hops[(size_t)i * N + j] = hik + hops[(size_t)k * N + j];     \
                            ++updates;                                                             \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
            pthread_barrier_wait(job->bar);                                                        \
        }                                                                                          \
        job->updates = updates;                                                                    \
        return NULL;                                                                               \
    }

FWO_DEFINE_WORKER_FAST(fwo_worker_fast_f64, double)
FWO_DEFINE_WORKER_FAST(fwo_worker_fast_f32, float)

static uint64_t fwo_relax_mt_fast(int32_t n, void *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                                  int32_t k_end, int32_t threads, void *(*worker)(void *))
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    fwo_fast_job job[256];
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads);
    for (int t = 0; t < threads; ++t) {
        job[t].n = n; job[t].k_begin = k_begin; job[t].k_end = k_end;
        job[t].tid = t; job[t].threads = threads;
        job[t].rate = rate; job[t].next = next; job[t].hops = hops; job[t].bar = &bar; job[t].updates = 0;
        pthread_create(&tid[t], NULL, worker, &job[t]);
    }
    uint64_t updates = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        updates += job[t].updates;
    }
    pthread_barrier_destroy(&bar);
    return updates;
}

uint64_t fwo_relax_mt_fast_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                               int32_t k_end, int32_t threads)
{
    return fwo_relax_mt_fast(n, rate, next, hops, k_begin, k_end, threads, fwo_worker_fast_f64);
}

uint64_t fwo_relax_mt_fast_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                               int32_t k_end, int32_t threads)
{
    return fwo_relax_mt_fast(n, rate, next, hops, k_begin, k_end, threads, fwo_worker_fast_f32);
}

/* ---------------------------------------------------------------------------------------------
 * The same loop TILED OVER PIVOTS, for the stretches of a solve at N = 32768 (every pivot step of the plain
 * loop streams the whole matrix through the host's memory: 256 pivots = a terabyte).  Row k and column k are
 * fixed points of step k (Algorithms.hs:50, :54), so a row can take several consecutive pivots in one visit
 * as long as each pivot row is used AS IT STOOD AT THE START OF ITS OWN STEP:
 *   1. the `tile` pivot rows of a tile are brought to their own time first -- row k takes the pivots
 *      bs .. k-1 of the tile, in order, from the snapshots already taken -- and each is snapshot then
 *      (rate, next and hops rows: the empty-ikPath rule reads next[k][j], the length hops[k][j]);
 *   2. every row i then takes the tile's pivots in ascending order from the snapshots (a pivot row only
 *      those after its own index; its own step does not touch it), while it is hot in the cache.
 * Per entry (i, j) the sequence of operands, products and strict compares is exactly the plain loop's: step
 * k reads r[i][k] as row i holds it after steps < k, and r[k][j] at time k.  Same results, same U;
 * tests/test_oracle_golden.py pins it to fwo_relax_* on every input kind, with and without next / hops,
 * ragged tiles and pivot ranges.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, k_begin, k_end, tid, threads, tile;
    void *rate, *w;                 /* w: tile x n snapshot rows of the rates */
    int32_t *next, *hops, *wn, *wh; /* wn / wh: tile x n snapshots of the pivots' next / hops rows */
    pthread_barrier_t *bar;
    uint64_t updates;
} fwo_tiled_job;

#define FWO_DEFINE_TILED(NAME, T)                                                                  \
    /* pivot k (snapshot rows wk / wnk / whk) onto row i; returns the number of updates */         \
    __attribute__((target_clones("avx2", "default"))) static uint64_t NAME##_row(                  \
        T *ri, int32_t *ni, int32_t *hi, int32_t n, int32_t i, int32_t k, const T *wk,             \
        const int32_t *wnk, const int32_t *whk)                                                    \
    {                                                                                              \
        uint64_t updates = 0;                                                                      \
        const T rik = ri[k];                                                                       \
        const int32_t nik = ni ? ni[k] : 0;                                                        \
        const int32_t hik = hi ? hi[k] : 0;                                                        \
        for (int32_t j0 = 0; j0 < n; j0 += FWO_CHUNK) {                                            \
            const int32_t j1 = j0 + FWO_CHUNK < n ? j0 + FWO_CHUNK : n;                            \
            int any = 0;                                                                           \
            for (int32_t j = j0; j < j1; ++j) any |= ri[j] < rik * wk[j];                          \
            if (!any) continue;                                                                    \
            for (int32_t j = j0; j < j1; ++j) {                                                    \
                if (j == i || j == k) continue;                                                    \
                const T c = rik * wk[j];                                                           \
                if (ri[j] < c) {                                                                   \
                    ri[j] = c;                                                                     \
                    if (ni) ni[j] = nik >= 0 ? nik : wnk[j];                                       \
                    if (hi) hi[j] = hik + whk[j];                                                  \
                    ++updates;                                                                     \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        return updates;                                                                            \
    }                                                                                              \
    static void *NAME(void *arg)                                                                   \
    {                                                                                              \
        fwo_tiled_job *job = (fwo_tiled_job *)arg;                                                 \
        const int32_t n = job->n;                                                                  \
        const size_t N = (size_t)n;                                                                \
        T *rate = (T *)job->rate, *w = (T *)job->w;                                                \
        int32_t *next = job->next, *hops = job->hops, *wn = job->wn, *wh = job->wh;                \
        const int32_t lo = (int32_t)(((int64_t)n * job->tid) / job->threads);                      \
        const int32_t hi_ = (int32_t)(((int64_t)n * (job->tid + 1)) / job->threads);               \
        uint64_t updates = 0;                                                                      \
        for (int32_t bs = job->k_begin; bs < job->k_end; bs += job->tile) {                        \
            const int32_t nb = bs + job->tile < job->k_end ? job->tile : job->k_end - bs;          \
            if (job->tid == 0) {                                                                   \
                for (int32_t t = 0; t < nb; ++t) {       /* 1. pivot rows to their own time */     \
                    const int32_t k = bs + t;                                                      \
                    T *rk = rate + (size_t)k * N;                                                  \
                    int32_t *nk = next ? next + (size_t)k * N : NULL;                              \
                    int32_t *hk = hops ? hops + (size_t)k * N : NULL;                              \
                    for (int32_t u = 0; u < t; ++u)                                                \
                        updates += NAME##_row(rk, nk, hk, n, k, bs + u, w + (size_t)u * N,         \
                                              wn ? wn + (size_t)u * N : NULL,                      \
                                              wh ? wh + (size_t)u * N : NULL);                     \
                    memcpy(w + (size_t)t * N, rk, N * sizeof(T));                                  \
                    if (nk) memcpy(wn + (size_t)t * N, nk, N * sizeof(int32_t));                   \
                    if (hk) memcpy(wh + (size_t)t * N, hk, N * sizeof(int32_t));                   \
                }                                                                                  \
            }                                                                                      \
            pthread_barrier_wait(job->bar);                                                        \
            for (int32_t i = lo; i < hi_; ++i) {         /* 2. every row, the tile's pivots */     \
                T *ri = rate + (size_t)i * N;                                                      \
                int32_t *ni = next ? next + (size_t)i * N : NULL;                                  \
                int32_t *hi2 = hops ? hops + (size_t)i * N : NULL;                                 \
                const int32_t first = (i >= bs && i < bs + nb) ? i - bs + 1 : 0;                   \
                for (int32_t u = first; u < nb; ++u)                                               \
                    updates += NAME##_row(ri, ni, hi2, n, i, bs + u, w + (size_t)u * N,            \
                                          wn ? wn + (size_t)u * N : NULL, wh ? wh + (size_t)u * N : NULL); \
            }                                                                                      \
            pthread_barrier_wait(job->bar);                                                        \
        }                                                                                          \
        job->updates = updates;                                                                    \
        return NULL;                                                                               \
    }

#include <stdlib.h>
#include <string.h>
FWO_DEFINE_TILED(fwo_tiled_f64, double)
FWO_DEFINE_TILED(fwo_tiled_f32, float)

static int64_t fwo_relax_mt_tiled(int32_t n, void *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                                  int32_t k_end, int32_t threads, int32_t tile, size_t es, void *(*worker)(void *))
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (tile < 1) tile = 1;
    if (n <= 0 || k_end <= k_begin) return 0;
    void *w = malloc((size_t)tile * n * es);
    int32_t *wn = next ? (int32_t *)malloc((size_t)tile * n * sizeof(int32_t)) : NULL;
    int32_t *wh = hops ? (int32_t *)malloc((size_t)tile * n * sizeof(int32_t)) : NULL;
    if (!w || (next && !wn) || (hops && !wh)) { free(w); free(wn); free(wh); return -1; }
    pthread_t tid[256];
    fwo_tiled_job job[256];
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads);
    for (int t = 0; t < threads; ++t) {
        job[t].n = n; job[t].k_begin = k_begin; job[t].k_end = k_end; job[t].tid = t; job[t].threads = threads;
        job[t].tile = tile; job[t].rate = rate; job[t].w = w; job[t].next = next; job[t].hops = hops;
        job[t].wn = wn; job[t].wh = wh; job[t].bar = &bar; job[t].updates = 0;
        pthread_create(&tid[t], NULL, worker, &job[t]);
    }
    int64_t updates = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        updates += (int64_t)job[t].updates;
    }
    pthread_barrier_destroy(&bar);
    free(w); free(wn); free(wh);
    return updates;
}

/* returns U, or -1 if the snapshot buffers could not be allocated */
int64_t fwo_relax_mt_tiled_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                               int32_t k_end, int32_t threads, int32_t tile)
{
    return fwo_relax_mt_tiled(n, rate, next, hops, k_begin, k_end, threads, tile, sizeof(double), fwo_tiled_f64);
}

int64_t fwo_relax_mt_tiled_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                               int32_t k_end, int32_t threads, int32_t tile)
{
    return fwo_relax_mt_tiled(n, rate, next, hops, k_begin, k_end, threads, tile, sizeof(float), fwo_tiled_f32);
}
