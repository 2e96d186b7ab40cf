// Microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 from registers (calibrates asmc_flow.hip's roofline).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip ; run: ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC, int WITH_LDS>
__global__ __launch_bounds__(512) void k(float* out, int iters, const float* src) {
    extern __shared__ float sp[];
    for (int i = threadIdx.x; i < 28 * 1024; i += 512) sp[i] = src[i & 1023];
    __syncthreads();
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    float b = threadIdx.x * 1e-3f;
    const float4* Ap = reinterpret_cast<const float4*>(sp) + (threadIdx.x & 63);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            float4 a4;
            if (WITH_LDS)
                a4 = Ap[(size_t)((it & 3) * 16 + g) * 64];
            else
                a4 = make_float4(b, b + 1.f, b + 2.f, b + 3.f);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NACC, int WITH_LDS>
void run(const char* name, float* out, float* src) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<NACC, WITH_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 115 * 1024);
    k<NACC, WITH_LDS><<<grid, 512, 115 * 1024>>>(out, 10, src);
    hipEventRecord(e0);
    k<NACC, WITH_LDS><<<grid, 512, 115 * 1024>>>(out, iters, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 64.0 * NACC * 4096.0;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *src;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&src, 4096);
    hipMemset(src, 0, 4096);
    run<1, 0>("1 acc, regs", out, src);
    run<2, 0>("2 acc, regs", out, src);
    run<4, 0>("4 acc, regs", out, src);
    run<1, 1>("1 acc, A from LDS", out, src);
    run<2, 1>("2 acc, A from LDS", out, src);
    run<4, 1>("4 acc, A from LDS", out, src);
    return 0;
}
