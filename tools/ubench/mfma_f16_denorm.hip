// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 subnormal inputs?  (The split-fp16 flow keeps residuals there.)
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; i++) a[i] = (_Float16)a_val, b[i] = (_Float16)b_val;
    floatx16 c;
    for (int i = 0; i < 16; i++) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d;
    hipMalloc(&d, 4);
    const float cases[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /* 2^-20 */, 1.0f}, {1.0f, 9.5367431640625e-07f},
                              {5.9604644775390625e-08f /* 2^-24 */, 1.0f}, {6.103515625e-05f /* 2^-14 */, 1.0f},
                              {9.5367431640625e-07f, 9.5367431640625e-07f}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cs[0], cs[1], d);
        float h;
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> %g (expected %g)\n", cs[0], cs[1], h, 16.0 * (double)cs[0] * (double)cs[1]);
    }
    return 0;
}
