// Probe for an f64 fold whose pivot-COLUMN operands are wave-uniform scalars (s_load from the Ct panel) instead of
// LDS reads: a wave owns RR rows x 64 columns of a 64 x 64 tile, a thread RR rows of ONE column, so per pivot a
// wave reads one ds_read_b64 (its W values) and RR scalars -- 8 / RR bytes of LDS per relaxation instead of the 4 of
// fused_main_arg_f64's 4 x 4 register tile (which sits at the edge of the LDS pipe).  Times one 64-pivot pass over
// an N x N matrix (no panels, results not checked: operands are synthetic).
//   hipcc --offload-arch=gfx950 -O3 tools/f64_fold_probe.hip -o build/probe/f64_fold_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double fmx(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int RR, int MINW>
__global__ __launch_bounds__(64 * (64 / RR), MINW) void fold_sgpr(double *__restrict__ rate, const double *__restrict__ ct,
                                                                 const double *__restrict__ w, int n, int ld, int hot)
{
    constexpr int THREADS = 64 * (64 / RR);
    __shared__ double sW[64][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int idx = threadIdx.x; idx < 64 * 64; idx += THREADS)
        sW[idx >> 6][idx & 63] = w[(size_t)(idx >> 6) * n + blockIdx.x * 64 + (idx & 63)];
    __syncthreads();
    const int i0 = blockIdx.y * 64 + wave * RR;
    double x[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) x[r] = rate[(size_t)(i0 + r) * n + blockIdx.x * 64 + lane];
    const double *cp = hot ? ct : ct + i0;      // hot: every wave reads the same 64 x RR scalars (all scalar-cache hits)
#pragma unroll 2
    for (int t = 0; t < 64; ++t) {
        const double wv = sW[t][lane];
#pragma unroll
        for (int r = 0; r < RR; ++r) x[r] = fmx(x[r], cp[(size_t)t * ld + r] * wv);
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) rate[(size_t)(i0 + r) * n + blockIdx.x * 64 + lane] = x[r];
}

// the shipped mapping for comparison: 4 x 4 entries per thread, both operands from LDS
__global__ __launch_bounds__(256, 2) void fold_lds(double *__restrict__ rate, const double *__restrict__ ct,
                                                   const double *__restrict__ w, int n, int ld)
{
    typedef double V2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double sW[64][64];
    __shared__ __attribute__((aligned(16))) double sC[64][64];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        sW[idx >> 6][idx & 63] = w[(size_t)(idx >> 6) * n + blockIdx.x * 64 + (idx & 63)];
        sC[idx >> 6][idx & 63] = ct[(size_t)(idx >> 6) * ld + blockIdx.y * 64 + (idx & 63)];
    }
    __syncthreads();
    double x[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            x[r][e] = rate[(size_t)(blockIdx.y * 64 + ti * 4 + r) * n + blockIdx.x * 64 + (e < 2 ? tj * 2 + e : 32 + tj * 2 + e - 2)];
#pragma unroll 2
    for (int t = 0; t < 64; ++t) {
        double c[4], wv[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const V2 cv = *reinterpret_cast<const V2 *>(&sC[t][ti * 4 + q * 2]);
            c[q * 2] = cv[0]; c[q * 2 + 1] = cv[1];
            const V2 wq = *reinterpret_cast<const V2 *>(&sW[t][tj * 2 + q * 32]);
            wv[q * 2] = wq[0]; wv[q * 2 + 1] = wq[1];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[r][e] = fmx(x[r][e], c[r] * wv[e]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            rate[(size_t)(blockIdx.y * 64 + ti * 4 + r) * n + blockIdx.x * 64 + (e < 2 ? tj * 2 + e : 32 + tj * 2 + e - 2)] = x[r][e];
}

template <typename F> static int bench(const char *name, F launch, int n)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms / reps;
    // floor: N^2 * 64 relaxations, one v_mul_f64 + one v_max_f64 each at 4.35 cycles, 1024 SIMDs x 64 lanes, 2.33 GHz
    const double floor_ms = (double)n * n * 64 / 64 / 1024 * 8.7 / 2.33e6;
    printf("%-44s %.3f ms per 64-pivot pass = %.0f ms per solve of %d passes; %.2f of the f64 fold floor (%.3f ms)\n",
           name, per, per * (n / 64), n / 64, floor_ms / per, floor_ms);
    return 0;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    double *rate, *ct, *w;
    CK(hipMalloc(&rate, (size_t)n * n * 8)); CK(hipMalloc(&ct, (size_t)64 * n * 8)); CK(hipMalloc(&w, (size_t)64 * n * 8));
    std::vector<double> h((size_t)64 * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5 + 0.4 * ((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(ct, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(rate, 0, (size_t)n * n * 8));
    const dim3 g(n / 64, n / 64);
    if (bench("LDS operands, 4 x 4 per thread, 2 WG/CU", [&] { hipLaunchKernelGGL(fold_lds, g, dim3(256), 0, 0, rate, ct, w, n, n); }, n)) return 1;
    if (bench("scalar C, 8 rows/thread, 512 thr, >=2 WG/CU", [&] { hipLaunchKernelGGL((fold_sgpr<8, 2>), g, dim3(512), 0, 0, rate, ct, w, n, n, 0); }, n)) return 1;
    if (bench("  the same, all scalar loads hit (one 4 KB strip)", [&] { hipLaunchKernelGGL((fold_sgpr<8, 2>), g, dim3(512), 0, 0, rate, ct, w, n, 8, 1); }, n)) return 1;
    if (bench("scalar C, 16 rows/thread, 256 thr", [&] { hipLaunchKernelGGL((fold_sgpr<16, 2>), g, dim3(256), 0, 0, rate, ct, w, n, n, 0); }, n)) return 1;
    if (bench("scalar C, 4 rows/thread, 1024 thr", [&] { hipLaunchKernelGGL((fold_sgpr<4, 1>), g, dim3(1024), 0, 0, rate, ct, w, n, n, 0); }, n)) return 1;
    return 0;
}
