// Do the fp64 matrix pipe and the fp64 vector ALU of one SIMD overlap?  (config 5's mutation step, k_pcn_mm at d = 128, is
// 144 v_mfma_f64_16x16x4_f64 + ~2 900 vector instructions - most of them fp64 - per 16 particles and runs at the SUM of the two.)
// 512-thread blocks, one per CU: waves 0-3 (one per SIMD) issue fp64 MFMAs, waves 4-7 (their SIMD partners) vector work.
// Reported: time of each role alone, of both together, and "hidden" = the part of the shorter one that disappeared.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(x) x x x x x x x x
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
#define DMFMA4 "v_mfma_f64_16x16x4_f64 v[32:39], v[10:11], v[12:13], v[32:39]\nv_mfma_f64_16x16x4_f64 v[40:47], v[10:11], v[12:13], v[40:47]\nv_mfma_f64_16x16x4_f64 v[48:55], v[10:11], v[12:13], v[48:55]\nv_mfma_f64_16x16x4_f64 v[56:63], v[10:11], v[12:13], v[56:63]\n"
#define FMA64_8 "v_fma_f64 v[10:11], %1, %1, v[10:11]\nv_fma_f64 v[12:13], %1, %1, v[12:13]\nv_fma_f64 v[14:15], %1, %1, v[14:15]\nv_fma_f64 v[16:17], %1, %1, v[16:17]\nv_fma_f64 v[18:19], %1, %1, v[18:19]\nv_fma_f64 v[20:21], %1, %1, v[20:21]\nv_fma_f64 v[22:23], %1, %1, v[22:23]\nv_fma_f64 v[24:25], %1, %1, v[24:25]\n"
#define FMA32_8 "v_fma_f32 v10, %0, %0, v10\nv_fma_f32 v12, %0, %0, v12\nv_fma_f32 v14, %0, %0, v14\nv_fma_f32 v16, %0, %0, v16\nv_fma_f32 v18, %0, %0, v18\nv_fma_f32 v20, %0, %0, v20\nv_fma_f32 v22, %0, %0, v22\nv_fma_f32 v24, %0, %0, v24\n"
#define XOR_8 "v_xor_b32 v10, %0, v11\nv_xor_b32 v12, %0, v13\nv_xor_b32 v14, %0, v15\nv_xor_b32 v16, %0, v17\nv_xor_b32 v18, %0, v19\nv_xor_b32 v20, %0, v21\nv_xor_b32 v22, %0, v23\nv_xor_b32 v24, %0, v25\n"
template <int VK>
__global__ __launch_bounds__(512) void k(int mode, int rep, unsigned* out) {
    const int wave = threadIdx.x >> 6;
    unsigned a = threadIdx.x * 2654435761u + 12345u;
    double q = 1.0 + threadIdx.x * 1e-6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        for (int r = 0; r < rep; r++) asm volatile(R8(DMFMA4 DMFMA4) : "+v"(a), "+v"(q)::CLOB);  // 64 fp64 MFMAs
    } else {
        if (!(mode & 2)) return;
        for (int r = 0; r < rep * 16; r++) {  // 16 x 64 vector instructions per 64 MFMAs of the partner
            if (VK == 0) asm volatile(R8(FMA64_8) : "+v"(a), "+v"(q)::CLOB);
            if (VK == 1) asm volatile(R8(FMA32_8) : "+v"(a), "+v"(q)::CLOB);
            if (VK == 2) asm volatile(R8(XOR_8) : "+v"(a), "+v"(q)::CLOB);
        }
    }
    if (a == 0x12345678u && q == 3.0) out[threadIdx.x] = a;
}
template <int VK>
void run(const char* name, unsigned* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL((k<VK>), dim3(256), dim3(512), 0, 0, mode, 10, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<VK>), dim3(256), dim3(512), 0, 0, mode, 1000, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const float lo = ms[1] < ms[2] ? ms[1] : ms[2];
    printf("%-12s | fp64 MFMA only %.3f ms (%.1f cycles per MFMA at 2.1 GHz) | vector only %.3f ms | both %.3f ms | hidden %.2f of the shorter\n", name, ms[1],
           ms[1] * 1e-3 * 2.1e9 / 64000.0, ms[2], ms[3], (ms[1] + ms[2] - ms[3]) / lo);
}
int main() {
    unsigned* out;
    (void)hipMalloc(&out, 4096);
    printf("partner waves on one SIMD: 64 v_mfma_f64_16x16x4_f64 per 1024 vector instructions\n");
    run<0>("v_fma_f64", out);
    run<1>("v_fma_f32", out);
    run<2>("v_xor_b32", out);
    return 0;
}
