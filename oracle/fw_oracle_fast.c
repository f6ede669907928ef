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
