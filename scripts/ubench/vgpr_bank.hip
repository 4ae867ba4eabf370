// micro-benchmark: do VGPR source-operand bank conflicts cost issue cycles on gfx950?
// v_fma_f32 with three VGPR sources taken from (a) registers that are 4 apart (same bank if banks = reg % 4),
// (b) consecutive registers (different banks); same for a VOP2 v_add_f32 with two VGPR sources.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float r = threadIdx.x * 1e-3f;
    asm volatile(
        "v_mov_b32 v16, %1\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 0.5\n v_mov_b32 v19, 0.25\n"
        "v_mov_b32 v20, %1\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.25\n"
        "v_mov_b32 v24, %1\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.25\n"
        "v_mov_b32 v28, %1\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 0.5\n v_mov_b32 v31, 0.25\n"
        "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n"
        "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"
        : "=v"(r) : "v"(r) : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28",
          "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)  // fma, sources 4 apart: v16, v20, v24 / v17, v21, v25 ...
            asm volatile(REP8("v_fma_f32 v32, v16, v20, v24\n v_fma_f32 v33, v17, v21, v25\n v_fma_f32 v34, v18, v22, v26\n v_fma_f32 v35, v19, v23, v27\n"
                              "v_fma_f32 v36, v20, v24, v28\n v_fma_f32 v37, v21, v25, v29\n v_fma_f32 v38, v22, v26, v30\n v_fma_f32 v39, v23, v27, v31\n")
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (MODE == 1)  // fma, consecutive sources: v16, v17, v18 / v17, v18, v19 ...
            asm volatile(REP8("v_fma_f32 v32, v16, v17, v18\n v_fma_f32 v33, v17, v18, v19\n v_fma_f32 v34, v18, v19, v20\n v_fma_f32 v35, v19, v20, v21\n"
                              "v_fma_f32 v36, v20, v21, v22\n v_fma_f32 v37, v21, v22, v23\n v_fma_f32 v38, v22, v23, v24\n v_fma_f32 v39, v23, v24, v25\n")
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (MODE == 2)  // add, sources 4 apart
            asm volatile(REP8("v_add_f32 v32, v16, v20\n v_add_f32 v33, v17, v21\n v_add_f32 v34, v18, v22\n v_add_f32 v35, v19, v23\n"
                              "v_add_f32 v36, v20, v24\n v_add_f32 v37, v21, v25\n v_add_f32 v38, v22, v26\n v_add_f32 v39, v23, v27\n")
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (MODE == 3)  // add, consecutive sources
            asm volatile(REP8("v_add_f32 v32, v16, v17\n v_add_f32 v33, v17, v18\n v_add_f32 v34, v18, v19\n v_add_f32 v35, v19, v20\n"
                              "v_add_f32 v36, v20, v21\n v_add_f32 v37, v21, v22\n v_add_f32 v38, v22, v23\n v_add_f32 v39, v23, v24\n")
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (MODE == 4)  // fma with one SGPR-free literal-free VGPR source repeated (v16, v16, v17): fewer distinct reads
            asm volatile(REP8("v_fma_f32 v32, v16, v16, v17\n v_fma_f32 v33, v17, v17, v18\n v_fma_f32 v34, v18, v18, v19\n v_fma_f32 v35, v19, v19, v20\n"
                              "v_fma_f32 v36, v20, v20, v21\n v_fma_f32 v37, v21, v21, v22\n v_fma_f32 v38, v22, v22, v23\n v_fma_f32 v39, v23, v23, v24\n")
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
        if (MODE == 5)  // fmac (VOP2: dst is the third source), sources 4 apart
            asm volatile(REP8("v_fmac_f32 v32, v16, v20\n v_fmac_f32 v33, v17, v21\n v_fmac_f32 v34, v18, v22\n v_fmac_f32 v35, v19, v23\n"
                              "v_fmac_f32 v36, v20, v24\n v_fmac_f32 v37, v21, v25\n v_fmac_f32 v38, v22, v26\n v_fmac_f32 v39, v23, v27\n")
                         ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
    }
    float s;
    asm volatile("v_add_f32 %0, v32, v33\n v_add_f32 %0, %0, v34\n v_add_f32 %0, %0, v36\n v_add_f32 %0, %0, v39\n" : "=v"(s));
    out[blockIdx.x * 256 + threadIdx.x] = s + r;
}
template <int MODE>
static void run(float* d, hipEvent_t e0, hipEvent_t e1) {
    static const char* names[] = {"v_fma_f32, 3 VGPR sources 4 apart", "v_fma_f32, 3 consecutive VGPR sources", "v_add_f32, 2 VGPR sources 4 apart",
                                  "v_add_f32, 2 consecutive VGPR sources", "v_fma_f32 a*a+b (2 distinct VGPRs)", "v_fmac_f32 (VOP2), sources 4 apart"};
    const int iters = 512, blocks = 4096;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts = (double)blocks * 4 * iters * 64;
    printf("%-45s %7.3f ms  %7.1f G wave-inst/s  %.2f cycles/inst/SIMD @2.4GHz\n", names[MODE], ms, insts / ms / 1e6,
           1024 * 2.4e9 / (insts / (ms * 1e-3)));
    if constexpr (MODE < 5) run<MODE + 1>(d, e0, e1);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    run<0>(d, e0, e1);
    return 0;
}
