/* c_abi_consumer.c -- a plain C99 program that uses libfwx the way a foreign host would: only
 * include/fwx.h and include/fwx_host.h, plain pointers and sizes.  Built and run by
 * tests/test_gpu_c_consumer.py on the GPU box; also compiled (syntax only) on CPU to keep the
 * headers valid C.
 *
 * 1. the reference's 4-vertex graph (src/test/MockData.hs:47-57) through fwx_solve_f64 and the
 *    expectations of src/test/AlgorithmsTest.hs:72-75;
 * 2. the same rates through the host mirror (updateRates -> findBestRate), README.md:188-246;
 * 3. the same matrix through fwx_solve_multi_f64 (row-partitioned, one process);
 * 4. a 200-vertex matrix through fwx_matrix_create_part (one partition per process; here world = 1, so
 *    the host's exchange callback has nothing to move) against fwx_solve_f64, with the event timings. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fwx.h"
#include "fwx_host.h"

#define CHECK(cond)                                                                                \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            fprintf(stderr, "FAILED %s (line %d)\n", #cond, __LINE__);                             \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

/* The host's side of the one exchange a partitioned solve has (fwx.h fwx_exchange_fn): make the panel on
 * this rank hold rank `owner`'s copy, ordered on `stream`.  With one rank the panel is already in place; a
 * real host enqueues ncclBroadcast(w, w, count, type, owner, comm, stream) here. */
static int panels_seen = 0;
static int exchange(void *ctx, int32_t k0, int32_t bt, int32_t owner, void *w, int32_t *wh, int64_t count,
                    void *stream)
{
    (void)ctx; (void)k0; (void)wh; (void)stream;
    if (owner != 0 || !w || count != (int64_t)bt * 200) return -1;
    ++panels_seen;
    return 0;
}

int main(void)
{
    /* vertices in matrix order: (GDAX,BTC) (GDAX,USD) (KRAKEN,BTC) (KRAKEN,USD) */
    double rate[16] = {0.0, 1001.0, 1.0, 0.0,   0.0008, 0.0, 0.0, 1.0,
                       1.0, 0.0, 0.0, 1000.0,   0.0, 1.0, 0.0009, 0.0};
    int32_t next[16] = {-1, 1, 2, -1,   0, -1, -1, 3,   0, -1, -1, 3,   -1, 1, 2, -1};
    int32_t hops[16] = {0, 1, 1, 0,   1, 0, 0, 1,   1, 0, 0, 1,   0, 1, 1, 0};
    int32_t path[8];
    uint64_t updates = 0;
    fwx_opts opts;
    int rc, len;

    if (fwx_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    memset(&opts, 0, sizeof(opts));
    opts.struct_size = (uint32_t)sizeof(opts);
    opts.device = -1;
    opts.updates_out = &updates;
    rc = fwx_solve_f64(4, rate, next, hops, &opts);
    if (rc) fprintf(stderr, "fwx_solve_f64: %s\n", fwx_strerror(rc));
    CHECK(rc == FWX_OK);
    /* AlgorithmsTest.hs:72-75: [1][0] = (0.0009, [3,2,0]); [2][3] = (1001, [0,1,3]) */
    CHECK(rate[1 * 4 + 0] == 0.0009 && next[1 * 4 + 0] == 3 && hops[1 * 4 + 0] == 3);
    CHECK(rate[2 * 4 + 3] == 1001.0 && next[2 * 4 + 3] == 0 && hops[2 * 4 + 3] == 3);
    len = fwx_follow_path(4, next, 2, 3, path, 8);
    CHECK(len == 3 && path[0] == 0 && path[1] == 1 && path[2] == 3);
    CHECK(updates > 0);

    {
        /* 3. the same call with the whole node behind it (row e'): one process, P partitions.
         * Here P = 3 LOGICAL partitions of device 0 (a device may be listed more than once);
         * on an 8-GPU node the list is {0,...,7} and the panels travel on RCCL. */
        double r2[16] = {0.0, 1001.0, 1.0, 0.0,   0.0008, 0.0, 0.0, 1.0,
                         1.0, 0.0, 0.0, 1000.0,   0.0, 1.0, 0.0009, 0.0};
        int32_t n2[16] = {-1, 1, 2, -1,   0, -1, -1, 3,   0, -1, -1, 3,   -1, 1, 2, -1};
        const int32_t devices[3] = {0, 0, 0};
        int i;
        rc = fwx_solve_multi_f64(4, r2, n2, NULL, 3, devices, FWX_XCHG_AUTO, NULL);
        if (rc) fprintf(stderr, "fwx_solve_multi_f64: %s\n", fwx_strerror(rc));
        CHECK(rc == FWX_OK);
        for (i = 0; i < 16; ++i) CHECK(r2[i] == rate[i] && n2[i] == next[i]);
    }

    {
        /* 4. one partition per process: create_part, slab upload, the ranks'
         * domain vote (one rank here), solve, slab download */
        enum { N = 200 };
        static double a[N * N], b[N * N];
        static int32_t an[N * N], bn[N * N];
        fwx_matrix *m = NULL;
        fwx_multi_timing tm;
        int32_t bits = 0;
        int i, j;
        for (i = 0; i < N; ++i)
            for (j = 0; j < N; ++j) {
                a[i * N + j] = b[i * N + j] = i == j ? 0.0 : 0.05 + 0.95 * (double)((i * 131 + j * 71) % 997) / 997.0;
                an[i * N + j] = bn[i * N + j] = i == j ? -1 : j;
            }
        CHECK(fwx_solve_f64(N, a, an, NULL, NULL) == FWX_OK);
        CHECK(fwx_matrix_create_part(&m, N, FWX_F64, 1, 0, 0, 1, -1, exchange, NULL) == FWX_OK && m != NULL);
        CHECK(fwx_matrix_set_timing(m, 1) == FWX_OK);
        CHECK(fwx_matrix_upload(m, b, bn, NULL) == FWX_OK);
        CHECK(fwx_matrix_solve(m, NULL) == FWX_ERR_INVALID);          /* the domain vote comes first */
        CHECK(fwx_matrix_upload(m, b, bn, NULL) == FWX_OK);
        CHECK(fwx_matrix_domain_bits(m, &bits) == FWX_OK && bits == 3);
        CHECK(fwx_matrix_set_domain(m, bits) == FWX_OK);
        CHECK(fwx_matrix_solve(m, NULL) == FWX_OK);
        CHECK(fwx_matrix_download(m, b, bn, NULL) == FWX_OK);
        for (i = 0; i < N * N; ++i) CHECK(a[i] == b[i] && an[i] == bn[i]);
        CHECK(panels_seen == 4);                                       /* 200 pivots = 4 panels */
        memset(&tm, 0, sizeof(tm));
        tm.struct_size = (uint32_t)sizeof(tm);
        CHECK(fwx_matrix_get_timing(m, &tm) == FWX_OK && tm.steps == 4 && tm.bulk_us > 0.0f);
        CHECK(fwx_matrix_destroy(m) == FWX_OK);
    }

    {
        fwxh_session *s = NULL;
        char out[2048];
        CHECK(fwxh_session_create(&s, -1) == 0 && s != NULL);
        CHECK(fwxh_serve_line(s, "2017-11-01T09:42:23+00:00 KRAKEN BTC USD 1000.0 0.0009", out, sizeof(out)) >= 0);
        CHECK(fwxh_serve_line(s, "2017-11-01T09:43:23+00:00 GDAX BTC USD 1001.0 0.0008", out, sizeof(out)) >= 0);
        CHECK(fwxh_session_state(s) == FWXH_STATE_OUTSYNC);
        CHECK(fwxh_serve_line(s, "KRAKEN BTC GDAX USD", out, sizeof(out)) >= 0);
        CHECK(strstr(out, "BEST_RATES_BEGIN KRAKEN BTC GDAX USD 1001.0") == out);
        CHECK(strstr(out, "(KRAKEN, BTC)\n(GDAX, BTC)\n(GDAX, USD)\nBEST_RATES_END") != NULL);
        CHECK(fwxh_session_state(s) == FWXH_STATE_INSYNC);
        CHECK(fwxh_session_destroy(s) == 0);
    }
    printf("c_abi_consumer: OK (U = %llu)\n", (unsigned long long)updates);
    return 0;
}
