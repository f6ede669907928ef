/* ASan/UBSan driver for the oracle's C restatement (oracle/fw_oracle.c is compiled into it). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

uint64_t fwo_relax_f64(int32_t, double *, int32_t *, int32_t *, int32_t, int32_t);
uint64_t fwo_relax_f32(int32_t, float *, int32_t *, int32_t *, int32_t, int32_t);
uint64_t fwo_relax_mt_f32(int32_t, float *, int32_t *, int32_t *, int32_t, int32_t, int32_t);
uint64_t fwo_relax_mt_f64(int32_t, double *, int32_t *, int32_t *, int32_t, int32_t, int32_t);
int fwo_copy_per_k_f64(int32_t, double *, int32_t *, int32_t *);
int32_t fwo_follow_path(int32_t, const int32_t *, int32_t, int32_t, int32_t *, int32_t);
/* oracle/fw_oracle_fast.c: the chunk-pre-check twin and the pivot-tiled form */
uint64_t fwo_relax_mt_fast_f64(int32_t, double *, int32_t *, int32_t *, int32_t, int32_t, int32_t);
uint64_t fwo_relax_mt_fast_f32(int32_t, float *, int32_t *, int32_t *, int32_t, int32_t, int32_t);
int64_t fwo_relax_mt_tiled_f64(int32_t, double *, int32_t *, int32_t *, int32_t, int32_t, int32_t, int32_t);
int64_t fwo_relax_mt_tiled_f32(int32_t, float *, int32_t *, int32_t *, int32_t, int32_t, int32_t, int32_t);

int main(void)
{
    for (int n = 0; n <= 37; n += (n < 5 ? 1 : 8)) {
        size_t nn = (size_t)n * n;
        double *r = malloc((nn + 1) * sizeof(double)), *r2 = malloc((nn + 1) * sizeof(double));
        float *f = malloc((nn + 1) * sizeof(float)), *f2 = malloc((nn + 1) * sizeof(float));
        int32_t *nx = malloc((nn + 1) * 4), *hp = malloc((nn + 1) * 4), *nx2 = malloc((nn + 1) * 4),
                *hp2 = malloc((nn + 1) * 4), *path = malloc((n + 1) * 4);
        unsigned s = 12345u + n;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                s = s * 1664525u + 1013904223u;
                double v = i == j ? 0.0 : 0.05 + 0.95 * (s >> 8) / 16777216.0;
                if ((s & 63) == 0) v = 0.0;
                r[i * n + j] = r2[i * n + j] = v;
                f[i * n + j] = f2[i * n + j] = (float)v;
                nx[i * n + j] = nx2[i * n + j] = (v > 0) ? j : -1;
                hp[i * n + j] = hp2[i * n + j] = v > 0;
            }
        fwo_relax_f64(n, r, nx, hp, 0, n);
        fwo_copy_per_k_f64(n, r2, nx2, hp2);
        for (size_t q = 0; q < nn; ++q)
            if (r[q] != r2[q] || nx[q] != nx2[q] || hp[q] != hp2[q]) return 1;
        uint64_t u1 = fwo_relax_f32(n, f, NULL, NULL, 0, n);
        uint64_t u2 = fwo_relax_mt_f32(n, f2, NULL, NULL, 0, n, 3);
        if (u1 != u2) return 2;
        for (size_t q = 0; q < nn; ++q)
            if (f[q] != f2[q]) return 3;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (fwo_follow_path(n, nx, i, j, path, n) != hp[i * n + j]) return 4;
        /* the multi-threaded form with next AND hops (5 threads: ragged row split) against the
         * single-threaded result above */
        s = 12345u + n;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                s = s * 1664525u + 1013904223u;
                double v = i == j ? 0.0 : 0.05 + 0.95 * (s >> 8) / 16777216.0;
                if ((s & 63) == 0) v = 0.0;
                r2[i * n + j] = v;
                nx2[i * n + j] = (v > 0) ? j : -1;
                hp2[i * n + j] = v > 0;
            }
        fwo_relax_mt_f64(n, r2, nx2, hp2, 0, n, 5);
        for (size_t q = 0; q < nn; ++q)
            if (r[q] != r2[q] || nx[q] != nx2[q] || hp[q] != hp2[q]) return 5;
        /* the fast twins against the plain loop's result (r, nx, hp), f64 with next + hops: odd thread counts, a
         * tile that does not divide n */
        for (int variant = 0; variant < 2; ++variant) {
            s = 12345u + n;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    s = s * 1664525u + 1013904223u;
                    double v = i == j ? 0.0 : 0.05 + 0.95 * (s >> 8) / 16777216.0;
                    if ((s & 63) == 0) v = 0.0;
                    r2[i * n + j] = v;
                    nx2[i * n + j] = (v > 0) ? j : -1;
                    hp2[i * n + j] = v > 0;
                }
            if (variant == 0) fwo_relax_mt_fast_f64(n, r2, nx2, hp2, 0, n, 3);
            else if (fwo_relax_mt_tiled_f64(n, r2, nx2, hp2, 0, n, 3, 5) < 0) return 6;
            for (size_t q = 0; q < nn; ++q)
                if (r[q] != r2[q] || nx[q] != nx2[q] || hp[q] != hp2[q]) return 7 + variant;
        }
        free(r); free(r2); free(f); free(f2); free(nx); free(hp); free(nx2); free(hp2); free(path);
    }
    puts("oracle sanitize ok");
    return 0;
}
