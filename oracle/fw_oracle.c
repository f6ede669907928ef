/*
 * fw_oracle.c -- CPU ORACLE for the max-product Floyd-Warshall hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (floydwarshall_amd/) never links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's relaxation loop
 *     runAlgo      /root/reference/src/lib/Algorithms.hs:42-61
 * on the dense SoA layout of SURVEY.md section 8a (rate[n*n] row-major, next[n*n] int32 =
 * index of `head _path` or -1 for the empty path, hops[n*n] int32 = `length _path`).
 *
 * Parity pinning: the reference is Haskell and no GHC exists in this image, so the reference
 * itself cannot be run.  This restatement is pinned by the reference's own golden vectors
 * (src/test/AlgorithmsTest.hs:55-58 initial 4x4, :72-75 solved 4x4, :45-47/:62-64 empty) --
 * see tests/test_oracle_golden.py -- and, above N=4, by agreement with the independent
 * list-faithful restatement in oracle/list_faithful.py (which carries whole `_path` lists
 * exactly as Algorithms.hs:55 concatenates them).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math: one IEEE multiply and one
 * ordered strict compare per relaxation, as GHC's Double `*` and `<`).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * In-place dense k-i-j, one pivot range [k_begin, k_end).
 *
 * Algorithms.hs:44   k ascending, terminates at k == matrixSize
 * Algorithms.hs:50   row k is copied unchanged                       -> `if (i == k) continue`
 * Algorithms.hs:54   entries with j == i or j == k are unchanged     -> `if (j == i || j == k)`
 * Algorithms.hs:58-61 operands come from the matrix at the START of step k; since row k and
 *                    column k are fixed points of step k the in-place update reads the same
 *                    values (checked against fwo_copy_per_k_* below).
 * Algorithms.hs:55   update iff old < new (strict, false on NaN); path = ikPath ++ kjPath, so
 *                    head = head ikPath (next[i][k]) -- or, when ikPath is EMPTY (next[i][k] < 0),
 *                    head kjPath (next[k][j]; row k is a fixed point of step k) -- and
 *                    length = len ik + len kj.  The empty-ikPath case cannot arise from what the
 *                    reference's parser admits (rates > 0, Parsers.hs:40: a winning product then
 *                    has two positive factors, and a positive entry always has a path); it arises
 *                    with negative rates or an input whose next is -1 on a non-zero rate.
 * Returns U, the number of successful relaxations.
 * ------------------------------------------------------------------------------------------- */
#define FWO_DEFINE_RELAX(NAME, T)                                                              \
    uint64_t NAME(int32_t n, T *rate, int32_t *next, int32_t *hops, int32_t k_begin,          \
                  int32_t k_end)                                                               \
    {                                                                                          \
        uint64_t updates = 0;                                                                  \
        const size_t N = (size_t)n;                                                            \
        for (int32_t k = k_begin; k < k_end; ++k) {                                            \
            const T *rk = rate + (size_t)k * N;                                                \
            for (int32_t i = 0; i < n; ++i) {                                                  \
                if (i == k) continue;                                                          \
                T *ri = rate + (size_t)i * N;                                                  \
                const T rik = ri[k];                                                           \
                const int32_t nik = next ? next[(size_t)i * N + k] : 0;                        \
                const int32_t hik = hops ? hops[(size_t)i * N + k] : 0;                        \
                for (int32_t j = 0; j < n; ++j) {                                              \
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
        return updates;                                                                        \
    }

FWO_DEFINE_RELAX(fwo_relax_f64, double)
FWO_DEFINE_RELAX(fwo_relax_f32, float)

/* ---------------------------------------------------------------------------------------------
 * Literal "new matrix per k" form (Algorithms.hs:44: `newMatrix = indices <&> updateRow`):
 * every step reads only the previous matrix and writes a fresh one.  Used by the tests to show
 * the in-place form above is the same function.  O(n^2) extra memory; small n only.
 * ------------------------------------------------------------------------------------------- */
#define FWO_DEFINE_COPY(NAME, T)                                                               \
    int NAME(int32_t n, T *rate, int32_t *next, int32_t *hops)                                 \
    {                                                                                          \
        const size_t N = (size_t)n, NN = N * N;                                                \
        if (n == 0) return 0;                                                                  \
        T *r2 = (T *)malloc(NN * sizeof(T));                                                   \
        int32_t *n2 = (int32_t *)malloc(NN * sizeof(int32_t));                                 \
        int32_t *h2 = (int32_t *)malloc(NN * sizeof(int32_t));                                 \
        if (!r2 || !n2 || !h2) { free(r2); free(n2); free(h2); return -1; }                    \
        for (int32_t k = 0; k < n; ++k) {                                                      \
            for (int32_t i = 0; i < n; ++i)                                                    \
                for (int32_t j = 0; j < n; ++j) {                                              \
                    const size_t ij = i * N + j, ik = i * N + k, kj = k * N + j;               \
                    T r = rate[ij];                                                            \
                    int32_t nx = next[ij], hp = hops[ij];                                      \
                    if (i != k && j != i && j != k) {                                          \
                        const T c = rate[ik] * rate[kj];                                       \
                        if (r < c) {                                                   \
                            r = c;                                                     \
                            nx = next[ik] >= 0 ? next[ik] : next[kj];                  \
                            hp = hops[ik] + hops[kj];                                  \
                        }                                                              \
                    }                                                                          \
                    r2[ij] = r; n2[ij] = nx; h2[ij] = hp;                                      \
                }                                                                              \
            memcpy(rate, r2, NN * sizeof(T));                                                  \
            memcpy(next, n2, NN * sizeof(int32_t));                                            \
            memcpy(hops, h2, NN * sizeof(int32_t));                                            \
        }                                                                                      \
        free(r2); free(n2); free(h2);                                                          \
        return 0;                                                                              \
    }

FWO_DEFINE_COPY(fwo_copy_per_k_f64, double)
FWO_DEFINE_COPY(fwo_copy_per_k_f32, float)

/* ---------------------------------------------------------------------------------------------
 * Multi-threaded form for the CPU baseline: inside one pivot step the rows are independent
 * (row k and column k are read-only during step k), so rows are split over `threads` workers
 * with a barrier per k.  Results are identical to fwo_relax_* (same operands, same order per
 * entry).  Rates, optional next, optional hops (hops need next).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, k_begin, k_end, tid, threads;
    void *rate;
    int32_t *next;
    int32_t *hops;
    pthread_barrier_t *bar;
    uint64_t updates;
} fwo_job;

#define FWO_DEFINE_WORKER(NAME, T)                                                             \
    static void *NAME(void *arg)                                                               \
    {                                                                                          \
        fwo_job *job = (fwo_job *)arg;                                                         \
        const int32_t n = job->n;                                                              \
        const size_t N = (size_t)n;                                                            \
        T *rate = (T *)job->rate;                                                              \
        int32_t *next = job->next;                                                             \
        int32_t *hops = job->hops;                                                             \
        const int32_t lo = (int32_t)(((int64_t)n * job->tid) / job->threads);                  \
        const int32_t hi = (int32_t)(((int64_t)n * (job->tid + 1)) / job->threads);            \
        uint64_t updates = 0;                                                                  \
        for (int32_t k = job->k_begin; k < job->k_end; ++k) {                                  \
            const T *rk = rate + (size_t)k * N;                                                \
            for (int32_t i = lo; i < hi; ++i) {                                                \
                if (i == k) continue;                                                          \
                T *ri = rate + (size_t)i * N;                                                  \
                const T rik = ri[k];                                                           \
                const int32_t nik = next ? next[(size_t)i * N + k] : 0;                        \
                const int32_t hik = hops ? hops[(size_t)i * N + k] : 0;                        \
                for (int32_t j = 0; j < n; ++j) {                                              \
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
            pthread_barrier_wait(job->bar);                                                    \
        }                                                                                      \
        job->updates = updates;                                                                \
        return NULL;                                                                           \
    }

FWO_DEFINE_WORKER(fwo_worker_f64, double)
FWO_DEFINE_WORKER(fwo_worker_f32, float)

static uint64_t fwo_relax_mt(int32_t n, void *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                             int32_t k_end, int32_t threads, void *(*worker)(void *))
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    fwo_job job[256];
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

uint64_t fwo_relax_mt_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                          int32_t k_end, int32_t threads)
{
    return fwo_relax_mt(n, rate, next, hops, k_begin, k_end, threads, fwo_worker_f64);
}

uint64_t fwo_relax_mt_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, int32_t k_begin,
                          int32_t k_end, int32_t threads)
{
    return fwo_relax_mt(n, rate, next, hops, k_begin, k_end, threads, fwo_worker_f32);
}

/* ---------------------------------------------------------------------------------------------
 * Follow next-hops from src until dst: the index form of the `_path` list that `optimum`
 * returns (Algorithms.hs:74-75).  Returns the path length (vertices after src, dst included),
 * 0 when next[src][dst] == -1 (empty path), -1 if more than `cap`/n hops are needed (a cycle:
 * only possible with arbitrage inputs, SURVEY.md section 7 "Full _path equality").
 * ------------------------------------------------------------------------------------------- */
int32_t fwo_follow_path(int32_t n, const int32_t *next, int32_t src, int32_t dst, int32_t *out,
                        int32_t cap)
{
    const size_t N = (size_t)n;
    int32_t len = 0, cur = src;
    if (src < 0 || dst < 0 || src >= n || dst >= n) return -2;
    if (next[(size_t)src * N + dst] < 0) return 0;
    while (cur != dst || len == 0) {
        const int32_t nx = next[(size_t)cur * N + dst];
        if (nx < 0 || len >= n || len >= cap) return -1;
        out[len++] = nx;
        cur = nx;
    }
    return len;
}
