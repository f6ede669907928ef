// mfma_probe.hip -- can the K=1 f32 MFMA (v_mfma_f32_32x32x1_2b_f32) serve as an EXACT outer-product
// multiplier for the fused max-form kernel?  (1) output layout, (2) bit-exactness against the IEEE
// product on hostile operands (subnormals, inf, zero), (3) issue cost alone and interleaved with
// v_max3_f32.   Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o build/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float V32 __attribute__((ext_vector_type(32)));

__global__ void one_mfma(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    V32 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 32; ++r) d[r * 64 + l] = c[r];
}

template <int MODE>   // 0: MFMA only, 1: max3 only, 2: 4 MFMA + 64 max3 interleaved (one pivot pair of a 64x64 tile)
__global__ __launch_bounds__(256) void rate(float *out, unsigned long long *cyc, float seed)
{
    const int l = threadIdx.x & 63;
    float a0 = seed + l, b0 = 1.0f + 0.001f * l;
    V32 x0 = {}, x1 = {};
    for (int r = 0; r < 32; ++r) { x0[r] = seed + r; x1[r] = seed - r; }
    const unsigned long long w0 = wall_clock64(), t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int t = 0; t < 2048; ++t) {
        V32 z = {};
        V32 p0, p1, p2, p3;
        if (MODE != 1) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, b0, z, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x1f32(b0, a0, z, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_32x32x1f32(a0, a0, z, 0, 0, 0);
            p3 = __builtin_amdgcn_mfma_f32_32x32x1f32(b0, b0, z, 0, 0, 0);
        } else {
            p0 = x1; p1 = x0; p2 = x1; p3 = x0;
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                x0[r] = __builtin_fmaxf(__builtin_fmaxf(x0[r], p0[r]), p1[r]);
                x1[r] = __builtin_fmaxf(__builtin_fmaxf(x1[r], p2[r]), p3[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 32; r += 8) { x0[r] += p0[r] + p1[r]; x1[r] += p2[r] + p3[r]; }
        }
        a0 += 1.0f;                                  // operands change: nothing is loop-invariant
        asm volatile("" : "+v"(a0), "+v"(b0));
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float acc = 0;
    for (int r = 0; r < 32; ++r) acc += x0[r] + x1[r];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE> static void run_rate(const char *name, float *out, unsigned long long *cyc)
{
    for (int w = 1; w <= 2; ++w) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * blocks);
        CK(hipMemcpy(h.data(), cyc, 2 * blocks * 8, hipMemcpyDeviceToHost));
        double c = 0, wl = 0;
        for (int i = 0; i < blocks; ++i) { c += h[2 * i]; wl += h[2 * i + 1]; }
        const double ghz = c / wl * 0.1;
        const double ns_iter = ms * 1e6 / (2048.0 * w);   // per loop trip per SIMD (one wave per SIMD per WG)
        printf("%-44s %d wave/SIMD: %.1f ns = %.0f cycles per pivot pair of a 64x64 wave tile (%.2f GHz)\n", name, w,
               ns_iter, ns_iter * ghz, ghz);
    }
}

int main()
{
    // ---- (1) layout ---------------------------------------------------------------------------
    std::vector<float> a(64), b(64), d(32 * 64);
    for (int l = 0; l < 64; ++l) { a[l] = (float)(l + 1); b[l] = ldexpf(1.0f, l % 32) * (l < 32 ? 1.0f : 3.0f); }
    float *da, *db, *dd;
    CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 32 * 64 * 4));
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 32 * 64 * 4, hipMemcpyDeviceToHost));
    // decode: value = a[la] * b[lb] with la, lb in the same block (lanes 0-31 / 32-63)
    int la_of[32][64], lb_of[32][64];
    bool ok = true;
    for (int r = 0; r < 32; ++r)
        for (int l = 0; l < 64; ++l) {
            int found = 0;
            for (int blk = 0; blk < 2 && !found; ++blk)
                for (int i = 0; i < 32 && !found; ++i)
                    for (int j = 0; j < 32 && !found; ++j)
                        if (d[r * 64 + l] == a[32 * blk + i] * b[32 * blk + j]) { la_of[r][l] = 32 * blk + i; lb_of[r][l] = 32 * blk + j; found = 1; }
            if (!found) { ok = false; la_of[r][l] = lb_of[r][l] = -1; }
        }
    printf("layout decoded: %s\n", ok ? "yes" : "NO");
    for (int r = 0; r < 32; r += 1)
        printf("  vgpr %2d: lane 0 -> (A lane %2d, B lane %2d)   lane 1 -> (%2d,%2d)   lane 32 -> (%2d,%2d)   lane 63 -> (%2d,%2d)\n", r,
               la_of[r][0], lb_of[r][0], la_of[r][1], lb_of[r][1], la_of[r][32], lb_of[r][32], la_of[r][63], lb_of[r][63]);
    // ---- (2) exactness ------------------------------------------------------------------------
    const float pool[] = {0.0f, 1.0f, 0.5f, 3.0f, 1e-3f, 0.999f, 1.17549435e-38f, 2.9e-39f, 1e-45f, 3.0e38f, 1.7e38f,
                          INFINITY, 7.0f, 1.0000001f, 0.33333334f, 1e-20f, 1e-25f, 6e-20f, 123456.79f, 2.5e-7f};
    const int np = sizeof(pool) / sizeof(pool[0]);
    srand(7);
    long long bad = 0, total = 0, sub = 0;
    for (int rep = 0; rep < 400 && ok; ++rep) {
        for (int l = 0; l < 64; ++l) {
            const bool hostile = rep % 2 == 0;
            a[l] = hostile ? pool[rand() % np] : (float)(rand() / (double)RAND_MAX) * (rep % 4 == 1 ? 1e-19f : 1.0f);
            b[l] = hostile ? pool[rand() % np] : (float)(rand() / (double)RAND_MAX) * (rep % 4 == 1 ? 1e-19f : 4.0f);
        }
        CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
        CK(hipMemcpy(d.data(), dd, 32 * 64 * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < 32; ++r)
            for (int l = 0; l < 64; ++l) {
                volatile float want = a[la_of[r][l]] * b[lb_of[r][l]];      // IEEE RNE product on the host
                float w = want, g = d[r * 64 + l];
                ++total;
                if (w != 0.0f && fabsf(w) < 1.17549435e-38f) ++sub;
                const bool same = (isnan(w) && isnan(g)) || memcmp(&w, &g, 4) == 0;
                if (!same) {
                    if (bad < 10) printf("  MISMATCH a=%a b=%a mfma=%a mul=%a\n", a[la_of[r][l]], b[lb_of[r][l]], g, w);
                    ++bad;
                }
            }
    }
    printf("exactness: %lld products (%lld with a subnormal result), %lld differ from the IEEE product\n", total, sub, bad);
    // ---- (3) cost -------------------------------------------------------------------------------
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&cyc, 2 * 512 * 8));
    run_rate<0>("4 x v_mfma_f32_32x32x1_2b_f32", out, cyc);
    run_rate<1>("64 x v_max3_f32", out, cyc);
    run_rate<2>("4 x MFMA + 64 x v_max3_f32 (fold of the pair)", out, cyc);
    return 0;
}
