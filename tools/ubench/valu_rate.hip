// Issue cost of single vector-ALU instructions on gfx950, in SIMD cycles per wave64 instruction.
// Each kernel runs REP x 64 copies of one instruction on 8 independent destination registers (no chain shorter than 8
// instructions), one block per CU, WPS waves per SIMD; the time of the launch divided by (instructions per wave x WPS)
// is the issue cost when the pipe is saturated.  Calibration: v_xor_b32 is a full-rate instruction (4 cycles).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define X8(s, a) s(a, 0) s(a, 1) s(a, 2) s(a, 3) s(a, 4) s(a, 5) s(a, 6) s(a, 7)
#define X64(s, a) X8(s, a) X8(s, a) X8(s, a) X8(s, a) X8(s, a) X8(s, a) X8(s, a) X8(s, a)

// 32-bit destination d[i], sources a (32-bit), b (32-bit)
#define OP32(ins, i) asm volatile(ins " %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));
#define OP32_3(ins, i) asm volatile(ins " %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));
#define OP32_1(ins, i) asm volatile(ins " %0, %1" : "+v"(d[i]) : "v"(a));
#define OP64(ins, i) asm volatile(ins " %0, %1, %2" : "+v"(q[i]) : "v"(qa), "v"(qb));
#define OP64_3(ins, i) asm volatile(ins " %0, %1, %2, %0" : "+v"(q[i]) : "v"(qa), "v"(qb));
#define OP64_1(ins, i) asm volatile(ins " %0, %1" : "+v"(q[i]) : "v"(qa));

#define KERNEL(name, BODY)                                                        \
    __global__ __launch_bounds__(1024) void name(int rep, unsigned* out) {        \
        unsigned d[8], a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x9E3779B9u; \
        double q[8], qa = 1.0 + threadIdx.x * 1e-6, qb = 0.999999 - threadIdx.x * 1e-7; \
        for (int i = 0; i < 8; i++) d[i] = a + i, q[i] = qa + i;                  \
        for (int r = 0; r < rep; r++) { BODY }                                    \
        unsigned acc = 0;                                                         \
        for (int i = 0; i < 8; i++) acc ^= d[i] ^ (unsigned)(long long)q[i];      \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                           \
    }

#define S_XOR(a_, i) OP32("v_xor_b32", i)
#define S_ADDU(a_, i) OP32("v_add_u32", i)
#define S_MULLO(a_, i) OP32("v_mul_lo_u32", i)
#define S_MULHI(a_, i) OP32("v_mul_hi_u32", i)
#define S_MUL24(a_, i) OP32("v_mul_u32_u24", i)
#define S_MULHI24(a_, i) OP32("v_mul_hi_u32_u24", i)
#define S_MAD24(a_, i) OP32_3("v_mad_u32_u24", i)
#define S_MAD64(a_, i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a), "v"(b) : "vcc");
#define S_FMA32(a_, i) OP32_3("v_fma_f32", i)
#define S_PKFMA32(a_, i) OP64_3("v_pk_fma_f32", i)
#define S_PKMUL32(a_, i) OP64("v_pk_mul_f32", i)
#define S_ADD64(a_, i) OP64("v_add_f64", i)
#define S_MUL64(a_, i) OP64("v_mul_f64", i)
#define S_FMA64(a_, i) OP64_3("v_fma_f64", i)
#define S_LDEXP64(a_, i) asm volatile("v_ldexp_f64 %0, %1, %2" : "+v"(q[i]) : "v"(qa), "v"(a));
#define S_CVT64U(a_, i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(q[i]) : "v"(a));
#define S_CVT3264(a_, i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(d[i]) : "v"(qa));
#define S_RCP64(a_, i) OP64_1("v_rcp_f64", i)
#define S_RSQ64(a_, i) OP64_1("v_rsq_f64", i)
#define S_SQRT64(a_, i) OP64_1("v_sqrt_f64", i)
#define S_FREXPM64(a_, i) OP64_1("v_frexp_mant_f64", i)
#define S_RNDNE64(a_, i) OP64_1("v_rndne_f64", i)
#define S_CNDMASK(a_, i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(d[i]) : "v"(a), "v"(b) : "vcc");
#define S_MAXI32(a_, i) OP32("v_max_i32", i)
#define S_MAX3F32(a_, i) OP32_3("v_max3_f32", i)
#define S_CVTPK(a_, i) OP32("v_cvt_pk_f16_f32", i)
#define S_MIXLO(a_, i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(d[i]) : "v"(a), "v"(b));
#define S_EXP32(a_, i) OP32_1("v_exp_f32", i)
#define S_LOG32(a_, i) OP32_1("v_log_f32", i)
#define S_RCP32(a_, i) OP32_1("v_rcp_f32", i)
#define S_SIN32(a_, i) OP32_1("v_sin_f32", i)
#define S_SQRT32(a_, i) OP32_1("v_sqrt_f32", i)
#define S_SWAP32(a_, i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(d[i]), "+v"(d[(i + 4) & 7]));
#define S_ALIGNBIT(a_, i) asm volatile("v_alignbit_b32 %0, %1, %1, 13" : "+v"(d[i]) : "v"(a));
#define S_PKMAXF16(a_, i) OP32("v_pk_max_f16", i)
#define S_PKMAX3F16(a_, i) OP32_3("v_pk_maximum3_f16", i)
#define S_MOV64(a_, i) OP64_1("v_mov_b64", i)
#define S_LSHLADD64(a_, i) asm volatile("v_lshl_add_u64 %0, %1, 3, %2" : "+v"(q[i]) : "v"(qa), "v"(qb));
#define S_READLANE(a_, i) { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(d[i])); asm volatile("" ::"s"(s_)); }
#define S_XOR3(a_, i) OP32_3("v_xor3_b32", i)
#define S_ADD3(a_, i) OP32_3("v_add3_u32", i)
#define S_DOT2(a_, i) OP32_3("v_dot2_f32_f16", i)
#define S_CVT64I(a_, i) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(q[i]) : "v"(a));
#define S_CMP64(a_, i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(q[i]), "v"(qa) : "vcc");

KERNEL(k_xor, X64(S_XOR, 0))
KERNEL(k_addu, X64(S_ADDU, 0))
KERNEL(k_mullo, X64(S_MULLO, 0))
KERNEL(k_mulhi, X64(S_MULHI, 0))
KERNEL(k_mul24, X64(S_MUL24, 0))
KERNEL(k_mulhi24, X64(S_MULHI24, 0))
KERNEL(k_mad24, X64(S_MAD24, 0))
KERNEL(k_mad64, X64(S_MAD64, 0))
KERNEL(k_fma32, X64(S_FMA32, 0))
KERNEL(k_pkfma32, X64(S_PKFMA32, 0))
KERNEL(k_pkmul32, X64(S_PKMUL32, 0))
KERNEL(k_add64, X64(S_ADD64, 0))
KERNEL(k_mul64, X64(S_MUL64, 0))
KERNEL(k_fma64, X64(S_FMA64, 0))
KERNEL(k_ldexp64, X64(S_LDEXP64, 0))
KERNEL(k_cvt64u, X64(S_CVT64U, 0))
KERNEL(k_cvt64i, X64(S_CVT64I, 0))
KERNEL(k_cvt3264, X64(S_CVT3264, 0))
KERNEL(k_rcp64, X64(S_RCP64, 0))
KERNEL(k_rsq64, X64(S_RSQ64, 0))
KERNEL(k_sqrt64, X64(S_SQRT64, 0))
KERNEL(k_frexpm64, X64(S_FREXPM64, 0))
KERNEL(k_rndne64, X64(S_RNDNE64, 0))
KERNEL(k_cmp64, X64(S_CMP64, 0))
KERNEL(k_cndmask, X64(S_CNDMASK, 0))
KERNEL(k_maxi32, X64(S_MAXI32, 0))
KERNEL(k_max3f32, X64(S_MAX3F32, 0))
KERNEL(k_cvtpk, X64(S_CVTPK, 0))
KERNEL(k_mixlo, X64(S_MIXLO, 0))
KERNEL(k_exp32, X64(S_EXP32, 0))
KERNEL(k_log32, X64(S_LOG32, 0))
KERNEL(k_rcp32, X64(S_RCP32, 0))
KERNEL(k_sin32, X64(S_SIN32, 0))
KERNEL(k_sqrt32, X64(S_SQRT32, 0))
KERNEL(k_swap32, X64(S_SWAP32, 0))
KERNEL(k_alignbit, X64(S_ALIGNBIT, 0))
KERNEL(k_pkmaxf16, X64(S_PKMAXF16, 0))
KERNEL(k_pkmax3f16, X64(S_PKMAX3F16, 0))
KERNEL(k_mov64, X64(S_MOV64, 0))
KERNEL(k_lshladd64, X64(S_LSHLADD64, 0))
KERNEL(k_readlane, X64(S_READLANE, 0))

KERNEL(k_add3, X64(S_ADD3, 0))
KERNEL(k_dot2, X64(S_DOT2, 0))

typedef void (*kern_t)(int, unsigned*);
struct Entry {
    const char* name;
    kern_t k;
};

int main(int argc, char** argv) {
    Entry tab[] = {{"v_xor_b32", k_xor}, {"v_add_u32", k_addu}, {"v_add3_u32", k_add3},
                   {"v_alignbit_b32", k_alignbit}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi},
                   {"v_mul_u32_u24", k_mul24}, {"v_mul_hi_u32_u24", k_mulhi24}, {"v_mad_u32_u24", k_mad24},
                   {"v_mad_u64_u32", k_mad64}, {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma32}, {"v_pk_mul_f32", k_pkmul32},
                   {"v_add_f64", k_add64}, {"v_mul_f64", k_mul64}, {"v_fma_f64", k_fma64}, {"v_ldexp_f64", k_ldexp64},
                   {"v_cvt_f64_u32", k_cvt64u}, {"v_cvt_f64_i32", k_cvt64i}, {"v_cvt_f32_f64", k_cvt3264}, {"v_rcp_f64", k_rcp64},
                   {"v_rsq_f64", k_rsq64}, {"v_sqrt_f64", k_sqrt64}, {"v_frexp_mant_f64", k_frexpm64}, {"v_rndne_f64", k_rndne64},
                   {"v_cmp_gt_f64", k_cmp64}, {"v_cndmask_b32", k_cndmask}, {"v_max_i32", k_maxi32}, {"v_max3_f32", k_max3f32},
                   {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_fma_mixlo_f16", k_mixlo}, {"v_exp_f32", k_exp32}, {"v_log_f32", k_log32},
                   {"v_rcp_f32", k_rcp32}, {"v_sin_f32", k_sin32}, {"v_sqrt_f32", k_sqrt32}, {"v_permlane32_swap", k_swap32},
                   {"v_pk_max_f16", k_pkmaxf16}, {"v_pk_maximum3_f16", k_pkmax3f16}, {"v_mov_b64", k_mov64},
                   {"v_lshl_add_u64", k_lshladd64}, {"v_readlane_b32", k_readlane}, {"v_dot2_f32_f16", k_dot2}};
    unsigned* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int rep = 4000;
    double base[3] = {0, 0, 0};
    printf("%-20s %10s %10s %10s   (cycles per wave64 instruction at 1 / 2 / 4 waves per SIMD; v_xor_b32 = 4 by definition at 2 waves)\n",
           "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto& e : tab) {
        double cyc[3];
        for (int w = 0; w < 3; w++) {
            const int wps = 1 << w, threads = 256 * wps;
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, 10, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, rep, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            cyc[w] = (double)ms * 1e-3 / ((double)rep * 64 * wps);  // seconds per instruction issued on a SIMD
            if (!strcmp(e.name, "v_xor_b32")) base[w] = cyc[w];
        }
        const double unit = base[1] / 4.0;  // seconds per cycle, from v_xor at 2 waves per SIMD
        printf("%-20s %10.2f %10.2f %10.2f\n", e.name, cyc[0] / unit, cyc[1] / unit, cyc[2] / unit);
    }
    printf("implied clock from v_xor_b32 = 4 cycles: %.3f GHz\n", 4.0 / base[1] * 1e-9);
    return 0;
}
