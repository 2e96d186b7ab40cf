// Second pass of tools/ubench/valu_rate.hip: instruction sequences written as ONE asm block per loop body (the compiler
// inserts nothing between them), for the forms whose first measurement looked implausible (v_cndmask_b32) and for
// short dependent idioms of the noise code.  Same calibration (v_xor_b32 = 4 units at 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define R8(x) x x x x x x x x
#define KERNEL(name, ASM)                                                                       \
    __global__ __launch_bounds__(1024) void name(int rep, unsigned* out) {                     \
        unsigned a = threadIdx.x * 2654435761u + 12345u;                                        \
        double q = 1.0 + threadIdx.x * 1e-6;                                                    \
        for (int r = 0; r < rep; r++)                                                           \
            asm volatile(R8(ASM) : "+v"(a), "+v"(q)::"v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                         "v20", "v21", "v22", "v23", "v24", "v25", "vcc", "s20", "s21", "s22", "s23");                     \
        if (a == 0x12345678u && q == 3.0) out[threadIdx.x] = a;                                 \
    }
// 8 instructions per ASM, 64 per loop body
KERNEL(k_xor, "v_xor_b32 v10, %0, v11\nv_xor_b32 v12, %0, v13\nv_xor_b32 v14, %0, v15\nv_xor_b32 v16, %0, v17\nv_xor_b32 v18, %0, v19\nv_xor_b32 v20, %0, v21\nv_xor_b32 v22, %0, v23\nv_xor_b32 v24, %0, v25\n")
KERNEL(k_cnd32, "v_cndmask_b32_e32 v10, %0, v11, vcc\nv_cndmask_b32_e32 v12, %0, v13, vcc\nv_cndmask_b32_e32 v14, %0, v15, vcc\nv_cndmask_b32_e32 v16, %0, v17, vcc\nv_cndmask_b32_e32 v18, %0, v19, vcc\nv_cndmask_b32_e32 v20, %0, v21, vcc\nv_cndmask_b32_e32 v22, %0, v23, vcc\nv_cndmask_b32_e32 v24, %0, v25, vcc\n")
KERNEL(k_cnd64, "v_cndmask_b32_e64 v10, %0, v11, s[20:21]\nv_cndmask_b32_e64 v12, %0, v13, s[20:21]\nv_cndmask_b32_e64 v14, %0, v15, s[20:21]\nv_cndmask_b32_e64 v16, %0, v17, s[20:21]\nv_cndmask_b32_e64 v18, %0, v19, s[20:21]\nv_cndmask_b32_e64 v20, %0, v21, s[20:21]\nv_cndmask_b32_e64 v22, %0, v23, s[20:21]\nv_cndmask_b32_e64 v24, %0, v25, s[20:21]\n")
// compare + two selects (a 64-bit select), 4 cmp + 4 cnd... 2 groups of (cmp, cnd, cnd, xor)
KERNEL(k_cmpcnd, "v_cmp_gt_f64 vcc, %1, v[10:11]\nv_cndmask_b32_e32 v12, %0, v13, vcc\nv_cndmask_b32_e32 v14, %0, v15, vcc\nv_xor_b32 v16, %0, v17\nv_cmp_gt_f64 vcc, %1, v[18:19]\nv_cndmask_b32_e32 v20, %0, v21, vcc\nv_cndmask_b32_e32 v22, %0, v23, vcc\nv_xor_b32 v24, %0, v25\n")
KERNEL(k_cmpu32cnd, "v_cmp_eq_u32 vcc, %0, v10\nv_cndmask_b32_e32 v12, %0, v13, vcc\nv_cndmask_b32_e32 v14, %0, v15, vcc\nv_xor_b32 v16, %0, v17\nv_cmp_eq_u32 vcc, %0, v18\nv_cndmask_b32_e32 v20, %0, v21, vcc\nv_cndmask_b32_e32 v22, %0, v23, vcc\nv_xor_b32 v24, %0, v25\n")
KERNEL(k_fma64, "v_fma_f64 v[10:11], %1, %1, v[10:11]\nv_fma_f64 v[12:13], %1, %1, v[12:13]\nv_fma_f64 v[14:15], %1, %1, v[14:15]\nv_fma_f64 v[16:17], %1, %1, v[16:17]\nv_fma_f64 v[18:19], %1, %1, v[18:19]\nv_fma_f64 v[20:21], %1, %1, v[20:21]\nv_fma_f64 v[22:23], %1, %1, v[22:23]\nv_fma_f64 v[24:25], %1, %1, v[24:25]\n")
// dependent fp64 chain (a Horner polynomial): one chain
KERNEL(k_fma64dep, "v_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\nv_fma_f64 v[10:11], %1, v[10:11], v[12:13]\n")
// fp64 FMA with a literal-free SGPR operand pair, as the scalar-coefficient kernels use
KERNEL(k_fma64s, "v_fma_f64 v[10:11], s[20:21], %1, v[10:11]\nv_fma_f64 v[12:13], s[22:23], %1, v[12:13]\nv_fma_f64 v[14:15], s[20:21], %1, v[14:15]\nv_fma_f64 v[16:17], s[22:23], %1, v[16:17]\nv_fma_f64 v[18:19], s[20:21], %1, v[18:19]\nv_fma_f64 v[20:21], s[22:23], %1, v[20:21]\nv_fma_f64 v[22:23], s[20:21], %1, v[22:23]\nv_fma_f64 v[24:25], s[22:23], %1, v[24:25]\n")
KERNEL(k_mulhilo, "v_mul_hi_u32 v10, %0, v11\nv_mul_lo_u32 v12, %0, v11\nv_mul_hi_u32 v14, %0, v15\nv_mul_lo_u32 v16, %0, v15\nv_mul_hi_u32 v18, %0, v19\nv_mul_lo_u32 v20, %0, v19\nv_mul_hi_u32 v22, %0, v23\nv_mul_lo_u32 v24, %0, v23\n")
KERNEL(k_add32f, "v_add_f32 v10, %0, v11\nv_add_f32 v12, %0, v13\nv_add_f32 v14, %0, v15\nv_add_f32 v16, %0, v17\nv_add_f32 v18, %0, v19\nv_add_f32 v20, %0, v21\nv_add_f32 v22, %0, v23\nv_add_f32 v24, %0, v25\n")
KERNEL(k_mov32, "v_mov_b32 v10, %0\nv_mov_b32 v12, %0\nv_mov_b32 v14, %0\nv_mov_b32 v16, %0\nv_mov_b32 v18, %0\nv_mov_b32 v20, %0\nv_mov_b32 v22, %0\nv_mov_b32 v24, %0\n")
KERNEL(k_and32, "v_and_b32 v10, %0, v11\nv_and_b32 v12, %0, v13\nv_and_b32 v14, %0, v15\nv_and_b32 v16, %0, v17\nv_and_b32 v18, %0, v19\nv_and_b32 v20, %0, v21\nv_and_b32 v22, %0, v23\nv_and_b32 v24, %0, v25\n")
KERNEL(k_lshr32, "v_lshrrev_b32 v10, 5, %0\nv_lshrrev_b32 v12, 5, %0\nv_lshrrev_b32 v14, 5, %0\nv_lshrrev_b32 v16, 5, %0\nv_lshrrev_b32 v18, 5, %0\nv_lshrrev_b32 v20, 5, %0\nv_lshrrev_b32 v22, 5, %0\nv_lshrrev_b32 v24, 5, %0\n")
KERNEL(k_maxf32, "v_max_f32 v10, %0, v11\nv_max_f32 v12, %0, v13\nv_max_f32 v14, %0, v15\nv_max_f32 v16, %0, v17\nv_max_f32 v18, %0, v19\nv_max_f32 v20, %0, v21\nv_max_f32 v22, %0, v23\nv_max_f32 v24, %0, v25\n")
KERNEL(k_pkaddf16, "v_pk_add_f16 v10, %0, v11\nv_pk_add_f16 v12, %0, v13\nv_pk_add_f16 v14, %0, v15\nv_pk_add_f16 v16, %0, v17\nv_pk_add_f16 v18, %0, v19\nv_pk_add_f16 v20, %0, v21\nv_pk_add_f16 v22, %0, v23\nv_pk_add_f16 v24, %0, v25\n")
KERNEL(k_cvtpkrtz, "v_cvt_pkrtz_f16_f32 v10, %0, v11\nv_cvt_pkrtz_f16_f32 v12, %0, v13\nv_cvt_pkrtz_f16_f32 v14, %0, v15\nv_cvt_pkrtz_f16_f32 v16, %0, v17\nv_cvt_pkrtz_f16_f32 v18, %0, v19\nv_cvt_pkrtz_f16_f32 v20, %0, v21\nv_cvt_pkrtz_f16_f32 v22, %0, v23\nv_cvt_pkrtz_f16_f32 v24, %0, v25\n")
KERNEL(k_mfma16, "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[48:63], v[10:13], v[14:17], v[48:63]\nv_mfma_f32_32x32x16_f16 v[64:79], v[10:13], v[14:17], v[64:79]\nv_mfma_f32_32x32x16_f16 v[80:95], v[10:13], v[14:17], v[80:95]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[48:63], v[10:13], v[14:17], v[48:63]\nv_mfma_f32_32x32x16_f16 v[64:79], v[10:13], v[14:17], v[64:79]\nv_mfma_f32_32x32x16_f16 v[80:95], v[10:13], v[14:17], v[80:95]\n")
KERNEL(k_mfma16dep, "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\n")
// MFMA interleaved with independent VALU of the SAME wave: 2 MFMA + 6 xor
KERNEL(k_mfma_xor, "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_xor_b32 v18, %0, v19\nv_xor_b32 v20, %0, v21\nv_xor_b32 v22, %0, v23\nv_mfma_f32_32x32x16_f16 v[48:63], v[10:13], v[14:17], v[48:63]\nv_xor_b32 v24, %0, v25\nv_xor_b32 v18, %0, v19\nv_xor_b32 v20, %0, v21\n")
KERNEL(k_mfma_fma64, "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_fma_f64 v[18:19], %1, %1, v[18:19]\nv_fma_f64 v[20:21], %1, %1, v[20:21]\nv_fma_f64 v[22:23], %1, %1, v[22:23]\nv_mfma_f32_32x32x16_f16 v[48:63], v[10:13], v[14:17], v[48:63]\nv_fma_f64 v[24:25], %1, %1, v[24:25]\nv_fma_f64 v[18:19], %1, %1, v[18:19]\nv_fma_f64 v[20:21], %1, %1, v[20:21]\n")
typedef void (*kern_t)(int, unsigned*);
struct Entry { const char* name; kern_t k; };
int main() {
    Entry tab[] = {{"v_xor_b32", k_xor}, {"v_mov_b32", k_mov32}, {"v_and_b32", k_and32}, {"v_lshrrev_b32", k_lshr32}, {"v_add_f32", k_add32f}, {"v_max_f32", k_maxf32},
                   {"v_pk_add_f16", k_pkaddf16}, {"v_cvt_pkrtz_f16_f32", k_cvtpkrtz},
                   {"v_cndmask_b32_e32 vcc", k_cnd32}, {"v_cndmask_b32_e64 sgpr", k_cnd64}, {"cmp_f64+2cnd+xor (x2)", k_cmpcnd}, {"cmp_u32+2cnd+xor (x2)", k_cmpu32cnd},
                   {"v_fma_f64 indep", k_fma64}, {"v_fma_f64 1 chain", k_fma64dep}, {"v_fma_f64 sgpr src", k_fma64s}, {"mul_hi+mul_lo pairs", k_mulhilo},
                   {"mfma 32x32x16 f16 x4acc", k_mfma16}, {"mfma 32x32x16 f16 dep", k_mfma16dep}, {"2 mfma + 6 xor", k_mfma_xor}, {"2 mfma + 6 fma64", k_mfma_fma64}};
    unsigned* out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int rep = 4000;
    double base = 0;
    printf("%-28s %10s %10s %10s   (units per instruction, v_xor_b32 at 2 waves/SIMD = 4; for mixed rows: per instruction of the 8-instruction group)\n", "sequence", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto& e : tab) {
        double t[3];
        for (int w = 0; w < 3; w++) {
            const int wps = 1 << w, threads = 256 * wps;
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, 10, out);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, rep, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            t[w] = (double)ms * 1e-3 / ((double)rep * 64 * wps);
        }
        if (!strcmp(e.name, "v_xor_b32")) base = t[1] / 4.0;
        printf("%-28s %10.2f %10.2f %10.2f\n", e.name, t[0] / base, t[1] / base, t[2] / base);
    }
    printf("1 unit = %.4f ns\n", base * 1e9);
    return 0;
}
