// Cross-wave overlap of the matrix pipe and the vector ALU on one SIMD, without loop overhead or dependent chains in
// the way (a re-measurement of tools/ubench/mfma_valu_overlap.hip, whose vector loops were latency bound).
// 512-thread blocks, one per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 (their SIMD partners) run role B.
// Role bodies are single asm blocks of 64 instructions.  Reported: time of A alone, B alone, both; "hidden" = the part of
// the shorter one that disappeared (1.0 = perfect overlap, 0.0 = serialised).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(x) x x x x x x x x
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"
#define MFMA4 "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]\nv_mfma_f32_32x32x16_f16 v[48:63], v[10:13], v[14:17], v[48:63]\nv_mfma_f32_32x32x16_f16 v[64:79], v[10:13], v[14:17], v[64:79]\nv_mfma_f32_32x32x16_f16 v[80:95], v[10:13], v[14:17], v[80:95]\n"
#define FMA64_8 "v_fma_f64 v[10:11], %1, %1, v[10:11]\nv_fma_f64 v[12:13], %1, %1, v[12:13]\nv_fma_f64 v[14:15], %1, %1, v[14:15]\nv_fma_f64 v[16:17], %1, %1, v[16:17]\nv_fma_f64 v[18:19], %1, %1, v[18:19]\nv_fma_f64 v[20:21], %1, %1, v[20:21]\nv_fma_f64 v[22:23], %1, %1, v[22:23]\nv_fma_f64 v[24:25], %1, %1, v[24:25]\n"
#define XOR_8 "v_xor_b32 v10, %0, v11\nv_xor_b32 v12, %0, v13\nv_xor_b32 v14, %0, v15\nv_xor_b32 v16, %0, v17\nv_xor_b32 v18, %0, v19\nv_xor_b32 v20, %0, v21\nv_xor_b32 v22, %0, v23\nv_xor_b32 v24, %0, v25\n"
#define MUL_8 "v_mul_hi_u32 v10, %0, v11\nv_mul_lo_u32 v12, %0, v11\nv_mul_hi_u32 v14, %0, v15\nv_mul_lo_u32 v16, %0, v15\nv_mul_hi_u32 v18, %0, v19\nv_mul_lo_u32 v20, %0, v19\nv_mul_hi_u32 v22, %0, v23\nv_mul_lo_u32 v24, %0, v23\n"
#define FMA32_8 "v_fma_f32 v10, %0, %0, v10\nv_fma_f32 v12, %0, %0, v12\nv_fma_f32 v14, %0, %0, v14\nv_fma_f32 v16, %0, %0, v16\nv_fma_f32 v18, %0, %0, v18\nv_fma_f32 v20, %0, %0, v20\nv_fma_f32 v22, %0, %0, v22\nv_fma_f32 v24, %0, %0, v24\n"
#define MIX_8 "v_fma_mixlo_f16 v10, %0, -1.0, v11 op_sel_hi:[1,0,0]\nv_fma_mixlo_f16 v12, %0, -1.0, v13 op_sel_hi:[1,0,0]\nv_fma_mixlo_f16 v14, %0, -1.0, v15 op_sel_hi:[1,0,0]\nv_fma_mixlo_f16 v16, %0, -1.0, v17 op_sel_hi:[1,0,0]\nv_cvt_pk_f16_f32 v18, %0, v19\nv_cvt_pk_f16_f32 v20, %0, v21\nv_max_i32 v22, %0, v23\nv_max_i32 v24, %0, v25\n"

// mode bit 0: role A (MFMA) active, bit 1: role B active; VK selects B's instruction; PRIO: s_setprio of the MFMA wave
template <int VK, int PRIO>
__global__ __launch_bounds__(512) void k(int mode, int rep, unsigned* out) {
    const int wave = threadIdx.x >> 6;
    unsigned a = threadIdx.x * 2654435761u + 12345u;
    double q = 1.0 + threadIdx.x * 1e-6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        for (int r = 0; r < rep; r++) asm volatile(R8(MFMA4 MFMA4) : "+v"(a), "+v"(q)::CLOB);  // 64 MFMAs
    } else {
        if (!(mode & 2)) return;
        for (int r = 0; r < rep * 8; r++) {  // 8 x 64 vector instructions per 64 MFMAs of the partner
            if (VK == 0) asm volatile(R8(FMA64_8) : "+v"(a), "+v"(q)::CLOB);
            if (VK == 1) asm volatile(R8(XOR_8) : "+v"(a), "+v"(q)::CLOB);
            if (VK == 2) asm volatile(R8(MUL_8) : "+v"(a), "+v"(q)::CLOB);
            if (VK == 3) asm volatile(R8(FMA32_8) : "+v"(a), "+v"(q)::CLOB);
            if (VK == 4) asm volatile(R8(MIX_8) : "+v"(a), "+v"(q)::CLOB);
        }
    }
    if (a == 0x12345678u && q == 3.0) out[threadIdx.x] = a;
}
// same-wave interleave: per 2 MFMAs, NV vector instructions of kind fma64
template <int NV>
__global__ __launch_bounds__(512) void k_same(int rep, unsigned* out) {
    unsigned a = threadIdx.x * 2654435761u + 12345u;
    double q = 1.0 + threadIdx.x * 1e-6;
    for (int r = 0; r < rep; r++) {
        if (NV == 8) asm volatile(R8("v_mfma_f32_32x32x16_f16 v[32:47], v[26:29], v[26:29], v[32:47]\n" FMA64_8 "v_mfma_f32_32x32x16_f16 v[48:63], v[26:29], v[26:29], v[48:63]\n") : "+v"(a), "+v"(q)::CLOB);
        if (NV == 16) asm volatile(R8("v_mfma_f32_32x32x16_f16 v[32:47], v[26:29], v[26:29], v[32:47]\n" FMA64_8 "v_mfma_f32_32x32x16_f16 v[48:63], v[26:29], v[26:29], v[48:63]\n" FMA64_8) : "+v"(a), "+v"(q)::CLOB);
    }
    if (a == 0x12345678u && q == 3.0) out[threadIdx.x] = a;
}
template <int VK, int PRIO>
void run(const char* name, unsigned* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL((k<VK, PRIO>), dim3(256), dim3(512), 0, 0, mode, 10, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<VK, PRIO>), dim3(256), dim3(512), 0, 0, mode, 2000, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const float lo = ms[1] < ms[2] ? ms[1] : ms[2], hi = ms[1] < ms[2] ? ms[2] : ms[1];
    printf("%-26s prio %d | MFMA only %.3f ms | VALU only %.3f ms | both %.3f ms | hidden %.2f of the shorter\n", name, PRIO, ms[1], ms[2], ms[3],
           (ms[1] + ms[2] - ms[3]) / lo);
    (void)hi;
}
int main() {
    unsigned* out;
    (void)hipMalloc(&out, 4096);
    printf("partner waves on one SIMD: 64 fp16 MFMAs (32x32x16) per 512 vector instructions\n");
    run<0, 0>("v_fma_f64", out);
    run<0, 2>("v_fma_f64", out);
    run<1, 0>("v_xor_b32", out);
    run<2, 0>("v_mul_hi/lo_u32", out);
    run<3, 0>("v_fma_f32", out);
    run<4, 0>("fma_mix/cvt_pk/max_i32", out);
    run<4, 2>("fma_mix/cvt_pk/max_i32", out);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int nv = 8; nv <= 16; nv += 8) {
        float ms;
        if (nv == 8) hipLaunchKernelGGL(k_same<8>, dim3(256), dim3(256), 0, 0, 10, out); else hipLaunchKernelGGL(k_same<16>, dim3(256), dim3(256), 0, 0, 10, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        if (nv == 8) hipLaunchKernelGGL(k_same<8>, dim3(256), dim3(256), 0, 0, 2000, out); else hipLaunchKernelGGL(k_same<16>, dim3(256), dim3(256), 0, 0, 2000, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("same wave, 1 wave/SIMD: 2 MFMA + %d v_fma_f64 per group: %.3f ms for 2000 x 8 groups = %.1f ns per group\n", nv, ms, ms * 1e6 / 16000);
    }
    return 0;
}
