// vgpr_bank.hip -- does gfx950 pay for two VALU sources in one VGPR bank (register index mod 4)?
// Fixed registers in inline asm: v_mul_f32 with src0/src1 in the SAME bank vs different banks, and the
// max-form triple (2 x v_mul_f32 + v_max3_f32) with the operand banks as the compiler happened to
// allocate them in fused_main_max vs conflict-free.  Build: hipcc --offload-arch=gfx950 -O3 tools/vgpr_bank.hip -o build/vgpr_bank
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int TRIPS = 4096;

// 16 instructions per repetition; KIND selects the register pattern
template <int KIND> __global__ __launch_bounds__(256) void stream(float *out, float seed)
{
    // initialise v32..v95 through ordinary code paths is not possible for fixed registers: do it in asm
    asm volatile(
        "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n"
        "v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n"
        "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"
        "v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
        "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n"
        "v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n"
        "v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n"
        "v_mov_b32 v60, %0\n v_mov_b32 v61, %0\n v_mov_b32 v62, %0\n v_mov_b32 v63, %0\n"
        :: "v"(seed)
        : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47",
          "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
#pragma unroll 1
    for (int t = 0; t < TRIPS; ++t) {
        if (KIND == 0)   // v_mul d, a, b : a and b in DIFFERENT banks (and d in a third)
            asm volatile(
                "v_mul_f32 v32, v41, v50\n v_mul_f32 v33, v42, v51\n v_mul_f32 v34, v43, v48\n v_mul_f32 v35, v40, v49\n"
                "v_mul_f32 v36, v45, v54\n v_mul_f32 v37, v46, v55\n v_mul_f32 v38, v47, v52\n v_mul_f32 v39, v44, v53\n"
                "v_mul_f32 v32, v41, v50\n v_mul_f32 v33, v42, v51\n v_mul_f32 v34, v43, v48\n v_mul_f32 v35, v40, v49\n"
                "v_mul_f32 v36, v45, v54\n v_mul_f32 v37, v46, v55\n v_mul_f32 v38, v47, v52\n v_mul_f32 v39, v44, v53\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        if (KIND == 1)   // a and b in the SAME bank
            asm volatile(
                "v_mul_f32 v32, v41, v49\n v_mul_f32 v33, v42, v50\n v_mul_f32 v34, v43, v51\n v_mul_f32 v35, v40, v48\n"
                "v_mul_f32 v36, v45, v53\n v_mul_f32 v37, v46, v54\n v_mul_f32 v38, v47, v55\n v_mul_f32 v39, v44, v52\n"
                "v_mul_f32 v32, v41, v49\n v_mul_f32 v33, v42, v50\n v_mul_f32 v34, v43, v51\n v_mul_f32 v35, v40, v48\n"
                "v_mul_f32 v36, v45, v53\n v_mul_f32 v37, v46, v54\n v_mul_f32 v38, v47, v55\n v_mul_f32 v39, v44, v52\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        if (KIND == 2)   // v_max3 x, x, p, q : x, p, q in three DIFFERENT banks
            asm volatile(
                "v_max3_f32 v32, v32, v41, v50\n v_max3_f32 v33, v33, v42, v51\n v_max3_f32 v34, v34, v43, v48\n v_max3_f32 v35, v35, v40, v49\n"
                "v_max3_f32 v36, v36, v45, v54\n v_max3_f32 v37, v37, v46, v55\n v_max3_f32 v38, v38, v47, v52\n v_max3_f32 v39, v39, v44, v53\n"
                "v_max3_f32 v32, v32, v41, v50\n v_max3_f32 v33, v33, v42, v51\n v_max3_f32 v34, v34, v43, v48\n v_max3_f32 v35, v35, v40, v49\n"
                "v_max3_f32 v36, v36, v45, v54\n v_max3_f32 v37, v37, v46, v55\n v_max3_f32 v38, v38, v47, v52\n v_max3_f32 v39, v39, v44, v53\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        if (KIND == 3)   // x and p in the SAME bank (q elsewhere)
            asm volatile(
                "v_max3_f32 v32, v32, v40, v49\n v_max3_f32 v33, v33, v41, v50\n v_max3_f32 v34, v34, v42, v51\n v_max3_f32 v35, v35, v43, v48\n"
                "v_max3_f32 v36, v36, v44, v53\n v_max3_f32 v37, v37, v45, v54\n v_max3_f32 v38, v38, v46, v55\n v_max3_f32 v39, v39, v47, v52\n"
                "v_max3_f32 v32, v32, v40, v49\n v_max3_f32 v33, v33, v41, v50\n v_max3_f32 v34, v34, v42, v51\n v_max3_f32 v35, v35, v43, v48\n"
                "v_max3_f32 v36, v36, v44, v53\n v_max3_f32 v37, v37, v45, v54\n v_max3_f32 v38, v38, v46, v55\n v_max3_f32 v39, v39, v47, v52\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
        if (KIND == 4)   // all three of x, p, q in the SAME bank
            asm volatile(
                "v_max3_f32 v32, v32, v40, v48\n v_max3_f32 v33, v33, v41, v49\n v_max3_f32 v34, v34, v42, v50\n v_max3_f32 v35, v35, v43, v51\n"
                "v_max3_f32 v36, v36, v44, v52\n v_max3_f32 v37, v37, v45, v53\n v_max3_f32 v38, v38, v46, v54\n v_max3_f32 v39, v39, v47, v55\n"
                "v_max3_f32 v32, v32, v40, v48\n v_max3_f32 v33, v33, v41, v49\n v_max3_f32 v34, v34, v42, v50\n v_max3_f32 v35, v35, v43, v51\n"
                "v_max3_f32 v36, v36, v44, v52\n v_max3_f32 v37, v37, v45, v53\n v_max3_f32 v38, v38, v46, v54\n v_max3_f32 v39, v39, v47, v55\n"
                ::: "v32","v33","v34","v35","v36","v37","v38","v39");
    }
    float acc;
    asm volatile("v_add_f32 %0, v32, v39" : "=v"(acc));
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND> static void run(const char *name, float *out)
{
    for (int wg_per_cu = 2; wg_per_cu <= 4; ++wg_per_cu) {
        const int blocks = 256 * wg_per_cu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(stream<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)TRIPS * 16 * 8;
        printf("%-44s %d wave/SIMD: %.3f ns per instruction per SIMD (%.1f ms)\n", name, wg_per_cu,
               ms * 1e6 / (instr * wg_per_cu), ms);
    }
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 1024 * 256 * 4));
    run<0>("v_mul_f32 d,a,b  a,b in different banks", out);
    run<1>("v_mul_f32 d,a,b  a,b in the SAME bank", out);
    run<2>("v_max3_f32 x,x,p,q  three different banks", out);
    run<3>("v_max3_f32 x,x,p,q  x and p in one bank", out);
    run<4>("v_max3_f32 x,x,p,q  all in one bank", out);
    return 0;
}
