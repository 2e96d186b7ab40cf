// Do MFMAs (fp32-input v_mfma_f32_32x32x2_f32; fp16 v_mfma_f32_32x32x16_f16) and vector-ALU work of ANOTHER wave on the same SIMD overlap?
// 512-thread blocks, one per CU: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (their SIMD partners) issue VALU FMAs.
// Prints the time of {MFMA only, VALU only, both}: "both" ~ max means separate pipes, ~ sum means a shared datapath.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int VKIND, int MKIND>  // VKIND 0: v_fma_f64, 1: v_fma_f32, 2: integer mul/xor (Philox-like); MKIND 0: fp32-input MFMA, 1: fp16 MFMA
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = wave < 4;
    if (mfma_wave) {
        if (!(mode & 1)) return;
        floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
        if (MKIND == 0) {
            for (int i = 0; i < iters; i++) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
            }
        } else {
            half8 hx, hy;
            for (int j = 0; j < 8; j++) hx[j] = (_Float16)(x + j), hy[j] = (_Float16)(y - j);
            for (int i = 0; i < 2 * iters; i++) {  // 32 cycles each: the same matrix-pipe time as the fp32 form's loop
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hx, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hx, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hy, a3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(mode & 2)) return;
        if (VKIND == 0) {
            double p = threadIdx.x * 1e-3, q = 1.0000001, r0 = 0.1, r1 = 0.2, r2 = 0.3, r3 = 0.4;
            for (int i = 0; i < iters * 16; i++) {  // 4 x 64 cycles of MFMA per iteration vs 16 x 4 x 4 cycles of f64 FMA
                r0 = fma(r0, q, p), r1 = fma(r1, q, p), r2 = fma(r2, q, p), r3 = fma(r3, q, p);
            }
            out[blockIdx.x * 512 + threadIdx.x] = (float)(r0 + r1 + r2 + r3);
        } else if (VKIND == 1) {
            float p = threadIdx.x * 1e-3f, q = 1.0000001f, r0 = 0.1f, r1 = 0.2f, r2 = 0.3f, r3 = 0.4f;
            for (int i = 0; i < iters * 16; i++) {
                r0 = fmaf(r0, q, p), r1 = fmaf(r1, q, p), r2 = fmaf(r2, q, p), r3 = fmaf(r3, q, p);
            }
            out[blockIdx.x * 512 + threadIdx.x] = r0 + r1 + r2 + r3;
        } else {
            unsigned r0 = threadIdx.x, r1 = 77, r2 = 99, r3 = 1234567;
            for (int i = 0; i < iters * 16; i++) {
                r0 = __umulhi(r0, 0xD2511F53u) ^ r1, r1 = r1 * 0xCD9E8D57u + r0, r2 = __umulhi(r2, 0xCD9E8D57u) ^ r3, r3 = r3 * 0xD2511F53u + r2;
            }
            out[blockIdx.x * 512 + threadIdx.x] = (float)(r0 ^ r1 ^ r2 ^ r3);
        }
    }
}

template <int VKIND, int MKIND>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 20000;
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL((k<VKIND, MKIND>), dim3(256), dim3(512), 0, 0, mode, 100, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<VKIND, MKIND>), dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    printf("%-10s MFMA only %.3f ms | VALU only %.3f ms | both %.3f ms  (max %.3f, sum %.3f)\n", name, ms[1], ms[2], ms[3],
           ms[1] > ms[2] ? ms[1] : ms[2], ms[1] + ms[2]);
    hipFree(out);
}

int main() {
    printf("fp32-input MFMA (v_mfma_f32_32x32x2_f32) in waves 0-3:\n");
    run<0, 0>("fma_f64");
    run<1, 0>("fma_f32");
    run<2, 0>("int mul");
    printf("fp16 MFMA (v_mfma_f32_32x32x16_f16) in waves 0-3:\n");
    run<0, 1>("fma_f64");
    run<1, 1>("fma_f32");
    run<2, 1>("int mul");
    return 0;
}
