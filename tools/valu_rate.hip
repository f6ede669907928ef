// valu_rate.hip -- issue cost of the VALU instructions the fused kernels are made of, measured on
// the device: long unrolled streams per wave, 1..4 waves per SIMD, cycles from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o build/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REG = 16;      // independent accumulators per lane
constexpr int INNER = 64;    // repetitions of the REG-instruction group per loop trip
constexpr int TRIPS = 512;     // per launch; `valu_rate long` repeats launches back to back for ~0.3 s

template <int KIND>
__global__ __launch_bounds__(256) void stream(float *out, unsigned long long *cycles, float seed)
{
    float a[REG], b[REG], c[REG];
#pragma unroll
    for (int i = 0; i < REG; ++i) { a[i] = seed + i + threadIdx.x; b[i] = 1.0f + 0.001f * i; c[i] = 0.5f + i; }
    const float s0 = seed;                       // kernel argument: lives in an SGPR
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
        for (int r = 0; r < INNER; ++r) {
#pragma unroll
            for (int i = 0; i < REG; ++i) {
                if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (KIND == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
                if (KIND == 4) {   // the max-form triple: two products, one fold (REG/2 entries... 3 instr)
                    float p, q;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(b[i]), "v"(c[i]));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(c[i]), "v"(b[(i + 1) % REG]));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(p), "v"(q));
                }
                if (KIND == 5) {   // compare-form quadruple: product, compare, two selects
                    float p;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(b[i]), "v"(c[i]));
                    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc"
                                 : "+v"(a[i]), "+v"(p), "+v"(c[i]) : "v"(b[i]) : "vcc");
                }
                if (KIND == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[i & ~1]));
                if (KIND == 7) asm volatile("v_max_f64 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[i & ~1]));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&a[i & ~1]) : "v"(*(double *)&b[i & ~1]));
                if (KIND == 9) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 10) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (KIND == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
                if (KIND == 13) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == 15) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if (KIND == 16) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[(i + 1) % REG]));
                if (KIND == 17) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[(i + 1) % REG]));
                if (KIND == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 19) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                // f64 with three distinct register pairs, and the f64 max-form pair as the kernels issue it
                if (KIND == 20 && !(i & 1))
                    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(*(double *)&a[i]) : "v"(*(double *)&b[i]), "v"(*(double *)&c[(i + 2) % REG]));
                if (KIND == 21 && !(i & 1)) {
                    double p;
                    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(*(double *)&b[i]), "v"(*(double *)&c[(i + 2) % REG]));
                    asm volatile("v_max_f64 %0, %0, %1" : "+v"(*(double *)&a[i]) : "v"(p));
                }
                if (KIND == 22 && !(i & 1)) {   // 4 products first, then 4 folds (distance between producer and consumer)
                    if ((i & 7) == 0) {
                        double p0, p1, p2, p3;
                        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p0) : "v"(*(double *)&b[i]), "v"(*(double *)&c[i]));
                        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p1) : "v"(*(double *)&b[i + 2]), "v"(*(double *)&c[i]));
                        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p2) : "v"(*(double *)&b[i + 4]), "v"(*(double *)&c[i]));
                        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p3) : "v"(*(double *)&b[i + 6]), "v"(*(double *)&c[i]));
                        asm volatile("v_max_f64 %0, %0, %1" : "+v"(*(double *)&a[i]) : "v"(p0));
                        asm volatile("v_max_f64 %0, %0, %1" : "+v"(*(double *)&a[i + 2]) : "v"(p1));
                        asm volatile("v_max_f64 %0, %0, %1" : "+v"(*(double *)&a[i + 4]) : "v"(p2));
                        asm volatile("v_max_f64 %0, %0, %1" : "+v"(*(double *)&a[i + 6]) : "v"(p3));
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < REG; ++i) acc += a[i] + c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { cycles[2 * blockIdx.x] = t1 - t0; cycles[2 * blockIdx.x + 1] = w1 - w0; }
}

// Sustained form: `reps` launches back to back (no gaps to speak of), 3 workgroups per CU like the
// fused main kernels; the clock is read inside the LAST launch, the rate over all of them.
template <int KIND> static void run_long(const char *name, int per_group, float *out, unsigned long long *cyc,
                                         int wg_per_cu, int reps)
{
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), cyc, 2 * blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0, wall = 0;
    for (int i = 0; i < blocks; ++i) { avg += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    const double ghz = avg / wall * 0.1;
    const double instr = (double)TRIPS * INNER * REG * per_group * reps;
    const double ns = ms * 1e6 / (instr * wg_per_cu);
    printf("%-34s %d wave/SIMD, %3d launches = %.1f ms: %.3f ns per instruction per SIMD = %.2f cycles at %.3f GHz (clock of the last launch)\n",
           name, wg_per_cu, reps, ms, ns, ns * ghz, ghz);
}

template <int KIND> static void run(const char *name, int per_group, float *out, unsigned long long *cyc)
{
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {   // 256 threads = 1 wave per SIMD
        const int blocks = 256 * wg_per_cu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * blocks);
        CK(hipMemcpy(h.data(), cyc, 2 * blocks * 8, hipMemcpyDeviceToHost));
        double avg = 0, wall = 0;
        for (int i = 0; i < blocks; ++i) { avg += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
        avg /= blocks; wall /= blocks;
        const double instr = (double)TRIPS * INNER * REG * per_group;     // per wave
        // s_memrealtime ticks at 100 MHz: shader clock = ticks(s_memtime) / ticks(realtime) * 100 MHz
        const double ghz = avg / wall * 0.1;
        const double ns = ms * 1e6 / (instr * wg_per_cu);
        printf("%-30s %d wave/SIMD: %.2f ns per instruction per SIMD = %.2f cycles at the measured %.2f GHz (%.2f ms)\n",
               name, wg_per_cu, ns, ns * ghz, ghz, ms);
    }
}

int main(int argc, char **argv)
{
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 2 * 1024 * 8));
    if (argc > 1 && !strcmp(argv[1], "long")) {
        // solve-length bursts (0.2-0.4 s): what clock and what issue rate does the chip SUSTAIN on the
        // instruction mixes of the fused main kernels?
        run_long<4>("2 x v_mul_f32 + v_max3_f32", 3, out, cyc, 3, 60);
        run_long<4>("2 x v_mul_f32 + v_max3_f32", 3, out, cyc, 4, 40);
        run_long<21>("v_mul_f64 + v_max_f64 (3 pairs)", 1, out, cyc, 2, 100);
        run_long<21>("v_mul_f64 + v_max_f64 (3 pairs)", 1, out, cyc, 3, 70);
        run_long<5>("v_mul + v_cmp + 2 x v_cndmask", 4, out, cyc, 3, 40);
        return 0;
    }
    run<0>("v_mul_f32 (v,v)", 1, out, cyc);
    run<3>("v_mul_f32 (s,v)", 1, out, cyc);
    run<2>("v_max_f32", 1, out, cyc);
    run<1>("v_max3_f32", 1, out, cyc);
    run<4>("2 x v_mul_f32 + v_max3_f32", 3, out, cyc);
    run<5>("v_mul + v_cmp + 2 x v_cndmask", 4, out, cyc);
    run<8>("v_pk_mul_f32", 1, out, cyc);
    run<6>("v_mul_f64", 1, out, cyc);
    run<7>("v_max_f64", 1, out, cyc);
    run<9>("v_max_u32", 1, out, cyc);
    run<19>("v_max_i32", 1, out, cyc);
    run<10>("v_max3_u32", 1, out, cyc);
    run<11>("v_add_f32", 1, out, cyc);
    run<12>("v_fma_f32", 1, out, cyc);
    run<13>("v_min_f32", 1, out, cyc);
    run<18>("v_and_b32", 1, out, cyc);
    run<14>("v_cndmask_b32", 1, out, cyc);
    run<15>("v_cmp_lt_f32", 1, out, cyc);
    run<16>("v_mul_f32 d,b,c (3 regs)", 1, out, cyc);
    run<17>("v_max_f32 d,b,c (3 regs)", 1, out, cyc);
    // per_group below = f64 instructions per REG-sized group / REG (the loop body covers REG floats)
    run<20>("v_mul_f64 d,b,c (3 pairs) [x0.5]", 1, out, cyc);
    run<21>("v_mul_f64+v_max_f64 pair [x1.0]", 1, out, cyc);
    run<22>("4 x mul_f64 then 4 x max_f64 [x1.0]", 1, out, cyc);
    run<0>("v_mul_f32 (v,v) again", 1, out, cyc);
    return 0;
}
