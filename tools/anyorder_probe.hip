// Does hipExtAnyOrderLaunch (a dispatch packet without the barrier bit) let two launches of ONE stream overlap on
// this device?  (hip_ext.h says the flag is not supported on GFX9xx boards.)  Each launch is one workgroup that
// spins for ~200 us and records s_memrealtime (100 MHz) at its start and end.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o build/probe/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(unsigned long long *out, int slot, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memrealtime(); }
    if (threadIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = t; }
}

int main()
{
    unsigned long long *d, h[8];
    CK(hipMalloc(&d, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemsetAsync(d, 0, 64, s));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 0, 20000ull);                  // 200 us
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 1, 20000ull);
        else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, 20000ull);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 2, 2000ull);                   // ordinary again: after both?
        CK(hipGetLastError());
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
        printf("%s: A [0, %.1f] us, B [%.1f, %.1f] us, C [%.1f, %.1f] us -> B %s A; C after both: %s\n",
               mode ? "any-order B" : "ordinary B ", (h[1] - h[0]) / 100.0, ((double)h[2] - h[0]) / 100.0,
               ((double)h[3] - h[0]) / 100.0, ((double)h[4] - h[0]) / 100.0, ((double)h[5] - h[0]) / 100.0,
               h[2] < h[1] ? "OVERLAPS" : "follows", (h[4] >= h[1] && h[4] >= h[3]) ? "yes" : "NO");
    }
    return 0;
}
