// tune_relax.hip -- standalone sweep of relax_k launch configurations on one MI355X.
// Development tool (not part of libfwx): includes the kernel source directly so that every
// (NV, RPB, UNROLL) instantiation is available.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude \
//         -Ifloydwarshall_amd/csrc tools/tune_relax.hip -o gpurun_out/tune_relax && gpurun_out/tune_relax
#include "../floydwarshall_amd/csrc/fwx_kernels.hip"
#include "../floydwarshall_amd/csrc/fwx_fused.hip"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_uniform(float *a, size_t n2, int n, unsigned seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        float u = (x >> 8) * (1.0f / 16777216.0f);
        a[i] = (i / n == i % n) ? 0.0f : 1.0f - u * 0.95f;
    }
}

struct Cfg { const char *name; hipError_t (*fn)(const fwx::RelaxArgs<float> &, hipStream_t); };

template <int NV, int RPB, int UNROLL, int MINW = 1, bool NT = false>
static hipError_t run_cfg(const fwx::RelaxArgs<float> &a, hipStream_t s)
{
    return fwx::launch_relax_cfg<float, 4, NV, RPB, UNROLL, MINW, NT>(a, s);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const int warm = argc > 2 ? atoi(argv[2]) : 1024;
    const int per = argc > 3 ? atoi(argv[3]) : 48;
    const int rounds = argc > 4 ? atoi(argv[4]) : 3;
    const size_t n2 = (size_t)n * n;
    float *d;
    CK(hipMalloc(&d, n2 * sizeof(float)));
    hipLaunchKernelGGL(fill_uniform, dim3(4096), dim3(256), 0, 0, d, n2, n, 12345u);
    CK(hipDeviceSynchronize());

    if (argc > 5 && !strcmp(argv[5], "solve")) {
        // PMC probe mode: one full solve with the PRODUCTION launch path (fwx::launch_relax), so
        // that `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- build/tune_relax 16384 0 0 0 solve` sees
        // exactly the dispatches bench.py times (torch's bundled HIP runtime crashes under --pmc).
        fwx::RelaxArgs<float> a;
        a.rate = d; a.next = nullptr; a.hops = nullptr; a.phops = nullptr;
        a.rows = n; a.n = n; a.row0 = 0; a.updates = nullptr;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        const int kmax = argc > 6 ? atoi(argv[6]) : n;      // optional: only the first kmax pivots
        const int sync_every = argc > 7 ? atoi(argv[7]) : 0; // optional: drain the queue regularly
        for (int k = 0; k < kmax; ++k) {
            a.k = k; a.prow = d + (size_t)k * n; a.flip = k & 1;
            CK(fwx::launch_relax<float>(a, 0));
            if (sync_every && (k + 1) % sync_every == 0) CK(hipDeviceSynchronize());
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("solve n=%d: %.3f ms, %.1f us/launch, %.3e relax/s\n", n, ms, 1e3 * ms / n,
               (double)n * n * n / (ms * 1e-3));
        return 0;
    }

    if (argc > 5 && !strcmp(argv[5], "fused")) {
        // PMC probe mode for the fused engine: `passes` passes of 64 pivots (panel, colpanel, main
        // in max form), a device sync after every pass so that rocprofv3 --pmc survives.
        const int passes = argc > 6 ? atoi(argv[6]) : 16;
        float *w, *ct;
        CK(hipMalloc(&w, (size_t)64 * n * sizeof(float)));
        CK(hipMalloc(&ct, (size_t)64 * n * sizeof(float)));
        fwx::FusedArgs<float> a;
        a.rate = d; a.next = nullptr; a.rows = n; a.n = n; a.row0 = 0; a.w = w; a.ct = ct;
        a.cnt = nullptr; a.ct_ld = n; a.updates = nullptr; a.nonneg = true;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int p = 0; p < passes; ++p) {
            a.k0 = p * 64; a.bt = 64;
            CK(fwx::launch_fused_panel<float>(d + (size_t)a.k0 * n, n, a.k0, 64, w, 0));
            CK(fwx::launch_fused_relax<float>(a, 0));
            CK(hipDeviceSynchronize());
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fused n=%d: %d passes, %.1f us/pass (synchronised after each pass)\n", n, passes,
               1e3 * ms / passes);
        return 0;
    }

    std::vector<Cfg> cfgs = {
        {"NV1 RPB8 U8", run_cfg<1, 8, 8>},        {"NV1 RPB8 U8 w2", run_cfg<1, 8, 8, 2>},
        {"NV1 RPB8 U4", run_cfg<1, 8, 4>},        {"NV1 RPB4 U4", run_cfg<1, 4, 4>},
        {"NV1 RPB8 U4 w2", run_cfg<1, 8, 4, 2>},  {"NV1 RPB16 U16", run_cfg<1, 16, 16>},
        {"NV1 RPB12 U12", run_cfg<1, 12, 12>},    {"NV1 RPB4 U4 w2", run_cfg<1, 4, 4, 2>},
        {"NV2 RPB4 U4", run_cfg<2, 4, 4>},        {"NV2 RPB8 U8", run_cfg<2, 8, 8>},
        {"NV1 RPB2 U2", run_cfg<1, 2, 2>},        {"NV1 RPB6 U6", run_cfg<1, 6, 6>},
    };

    fwx::RelaxArgs<float> a;
    a.rate = d; a.next = nullptr; a.hops = nullptr; a.phops = nullptr;
    a.rows = n; a.n = n; a.row0 = 0; a.updates = nullptr;
    int k = 0;
    auto step = [&](const Cfg &c, int serp) {
        a.k = k % n; a.prow = d + (size_t)a.k * n; a.flip = serp ? (k & 1) : 0;
        CK(c.fn(a, 0));
        ++k;
    };
    for (int i = 0; i < warm; ++i) step(cfgs[0], 1);
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> best(cfgs.size() * 2, 1e30f), sum(cfgs.size() * 2, 0.f);
    for (int r = 0; r < rounds; ++r)
        for (size_t c = 0; c < cfgs.size(); ++c)
            for (int serp = 1; serp >= 0; --serp) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < per; ++i) step(cfgs[c], serp);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                float us = 1e3f * ms / per;
                if (us < best[c * 2 + serp]) best[c * 2 + serp] = us;
                sum[c * 2 + serp] += us;
            }
    printf("n=%d warm=%d per=%d rounds=%d  (us/launch min|mean, GB/s at min; serpentine on / off)\n", n, warm, per, rounds);
    const double bytes = (double)n2 * 4;
    for (size_t c = 0; c < cfgs.size(); ++c)
        printf("%-14s  serp: %7.1f | %7.1f us  %6.0f GB/s    noserp: %7.1f | %7.1f us  %6.0f GB/s\n", cfgs[c].name,
               best[c * 2 + 1], sum[c * 2 + 1] / rounds, bytes / best[c * 2 + 1] * 1e-3,
               best[c * 2], sum[c * 2] / rounds, bytes / best[c * 2] * 1e-3);
    return 0;
}
