// graph_probe.hip -- does replaying the per-k launch chain as a hipGraph shorten the launch-bound
// regime (n <= ~2048)?  Captures fwx_dev_relax (n launches) into a graph and times stream launches
// vs graph replay.  Build: hipcc --offload-arch=gfx950 -O2 -Iinclude tools/graph_probe.hip
//        -Lfloydwarshall_amd -lfwx -Wl,-rpath,$PWD/floydwarshall_amd -o build/graph_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#include "fwx.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int reps = 5;
    for (int ai = 1; ai < argc; ++ai) {
        const int n = atoi(argv[ai]);
        const size_t nn = (size_t)n * n;
        std::vector<double> h(nn);
        srand(1);
        for (size_t i = 0; i < nn; ++i) h[i] = 0.05 + 0.95 * (rand() / (double)RAND_MAX);
        for (int i = 0; i < n; ++i) h[(size_t)i * n + i] = 0.0;
        double *d, *d0; int32_t *nx, *hp;
        CK(hipMalloc(&d, nn * 8)); CK(hipMalloc(&d0, nn * 8));
        CK(hipMalloc(&nx, nn * 4)); CK(hipMalloc(&hp, nn * 4));
        CK(hipMemcpy(d0, h.data(), nn * 8, hipMemcpyHostToDevice));
        CK(hipMemset(nx, 0, nn * 4)); CK(hipMemset(hp, 0, nn * 4));
        hipStream_t s; CK(hipStreamCreate(&s));
        fwx_slab slab = {n, 0, n, FWX_F64, d, nx, hp};
        fwx_pivots piv = {0, n, d, hp, n};
        // stream launches
        double best_stream = 1e9, best_graph = 1e9, t_inst = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpyAsync(d, d0, nn * 8, hipMemcpyDeviceToDevice, s));
            CK(hipStreamSynchronize(s));
            const double t0 = now();
            if (fwx_dev_relax(&slab, &piv, 1, nullptr, s)) { printf("relax failed\n"); return 1; }
            CK(hipStreamSynchronize(s));
            best_stream = std::min(best_stream, now() - t0);
        }
        // capture
        hipGraph_t g; hipGraphExec_t ge;
        double t0 = now();
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        if (fwx_dev_relax(&slab, &piv, 1, nullptr, s)) { printf("relax (capture) failed\n"); return 1; }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        t_inst = now() - t0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemcpyAsync(d, d0, nn * 8, hipMemcpyDeviceToDevice, s));
            CK(hipStreamSynchronize(s));
            const double t1 = now();
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            best_graph = std::min(best_graph, now() - t1);
        }
        printf("n=%5d f64+next+hops: stream %.3f ms (%.2f us/launch), graph replay %.3f ms (%.2f us/launch), capture+instantiate %.3f ms\n",
               n, 1e3 * best_stream, 1e6 * best_stream / n, 1e3 * best_graph, 1e6 * best_graph / n, 1e3 * t_inst);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        CK(hipStreamDestroy(s));
        CK(hipFree(d)); CK(hipFree(d0)); CK(hipFree(nx)); CK(hipFree(hp));
    }
    return 0;
}
