// The fp32 -> (hi, lo) fp16-pair conversion of the split-fp16 flow layers (asmc_flow_dev.h split2_f16), two forms, per PAIR of
// values (VERDICT r3 item 4 asked for the numbers either way):
//   A (shipped): ReLU (2 v_max_i32), hi = v_cvt_pk_f16_f32 (round to nearest), lo = rn16(v - hi) as two v_fma_mix{lo,hi}_f16
//   B (truncation split): ReLU (2 v_max_i32), hi = v & 0xFFFFE000 (2 v_and_b32; 11 significant bits: exact in fp16),
//     lo = v - hi in fp32 (one v_pk_add_f32 with neg, or two v_sub_f32), two v_cvt_pk_f16_f32
// plus half a v_pk_maximum3_f16 (range check) in both.  Same calibration as valu_rate2.hip (v_xor_b32 = 4 units at 2 waves / SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define R4(x) x x x x
#define KERNEL(name, ASM)                                                                       \
    __global__ __launch_bounds__(1024) void name(int rep, unsigned* out) {                     \
        unsigned a = threadIdx.x * 2654435761u + 12345u;                                        \
        float f0 = 1.0f + threadIdx.x * 1e-3f, f1 = -0.5f + threadIdx.x * 1e-3f;                \
        for (int r = 0; r < rep; r++)                                                           \
            asm volatile(R4(ASM) : "+v"(a), "+v"(f0), "+v"(f1)::"v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                         "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "vcc");                                     \
        if (a == 0x12345678u && f0 == 3.0f) out[threadIdx.x] = a;                               \
    }
KERNEL(k_xor, "v_xor_b32 v10, %0, v11\nv_xor_b32 v12, %0, v13\nv_xor_b32 v14, %0, v15\nv_xor_b32 v16, %0, v17\nv_xor_b32 v18, %0, v19\nv_xor_b32 v20, %0, v21\nv_xor_b32 v22, %0, v23\nv_xor_b32 v24, %0, v25\n")
// two pairs per ASM (8 pairs per loop body)
#define PAIR_A(x0, x1, r0, r1, h, l)                                   \
    "v_max_i32 " r0 ", " x0 ", 0\nv_max_i32 " r1 ", " x1 ", 0\n"       \
    "v_cvt_pk_f16_f32 " h ", " r0 ", " r1 "\n"                         \
    "v_fma_mixlo_f16 " l ", " h ", -1.0, " r0 " op_sel_hi:[1,0,0]\n"   \
    "v_fma_mixhi_f16 " l ", " h ", -1.0, " r1 " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
KERNEL(k_formA, PAIR_A("%1", "%2", "v10", "v11", "v12", "v13") PAIR_A("%2", "%1", "v14", "v15", "v16", "v17")
                "v_pk_maximum3_f16 v18, v18, v12, v16\n")
#define PAIR_B(x0, x1, r01a, r01b, h01a, h01b, l01, h, l)                                   \
    "v_max_i32 " r01a ", " x0 ", 0\nv_max_i32 " r01b ", " x1 ", 0\n"                        \
    "v_and_b32 " h01a ", 0xffffe000, " r01a "\nv_and_b32 " h01b ", 0xffffe000, " r01b "\n"  \
    "v_pk_add_f32 " l01 ", " r01a ":" r01b "], " h01a ":" h01b "] neg_lo:[0,1] neg_hi:[0,1]\n"
// (register pairs spelled out below: the macro above documents the sequence)
KERNEL(k_formB, "v_max_i32 v10, %1, 0\nv_max_i32 v11, %2, 0\nv_and_b32 v12, 0xffffe000, v10\nv_and_b32 v13, 0xffffe000, v11\n"
                "v_pk_add_f32 v[14:15], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_cvt_pk_f16_f32 v16, v12, v13\nv_cvt_pk_f16_f32 v17, v14, v15\n"
                "v_max_i32 v20, %2, 0\nv_max_i32 v21, %1, 0\nv_and_b32 v22, 0xffffe000, v20\nv_and_b32 v23, 0xffffe000, v21\n"
                "v_pk_add_f32 v[24:25], v[20:21], v[22:23] neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_cvt_pk_f16_f32 v26, v22, v23\nv_cvt_pk_f16_f32 v27, v24, v25\n"
                "v_pk_maximum3_f16 v18, v18, v16, v26\n")
KERNEL(k_formB2, "v_max_i32 v10, %1, 0\nv_max_i32 v11, %2, 0\nv_and_b32 v12, 0xffffe000, v10\nv_and_b32 v13, 0xffffe000, v11\n"
                 "v_sub_f32 v14, v10, v12\nv_sub_f32 v15, v11, v13\n"
                 "v_cvt_pk_f16_f32 v16, v12, v13\nv_cvt_pk_f16_f32 v17, v14, v15\n"
                 "v_max_i32 v20, %2, 0\nv_max_i32 v21, %1, 0\nv_and_b32 v22, 0xffffe000, v20\nv_and_b32 v23, 0xffffe000, v21\n"
                 "v_sub_f32 v24, v20, v22\nv_sub_f32 v25, v21, v23\n"
                 "v_cvt_pk_f16_f32 v26, v22, v23\nv_cvt_pk_f16_f32 v27, v24, v25\n"
                 "v_pk_maximum3_f16 v18, v18, v16, v26\n")
typedef void (*kern_t)(int, unsigned*);
struct Entry { const char* name; kern_t k; double pairs; };
int main() {
    Entry tab[] = {{"v_xor_b32 (calibration)", k_xor, 8.0}, {"A: max,max,cvt_pk,mixlo,mixhi", k_formA, 2.0},
                   {"B: max,max,and,and,pk_add_f32,cvt,cvt", k_formB, 2.0}, {"B2: ... two v_sub_f32 ...", k_formB2, 2.0}};
    unsigned* out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int rep = 4000;
    double base = 0;
    printf("%-44s %10s %10s %10s   (units per PAIR of values incl. half a range check; v_xor_b32 at 2 waves/SIMD = 4 units per instruction)\n", "sequence", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
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
            t[w] = (double)ms * 1e-3 / ((double)rep * 4 * e.pairs * wps);  // per pair (per instruction for the calibration row)
        }
        if (!strncmp(e.name, "v_xor_b32", 9)) base = t[1] / 4.0;
        printf("%-44s %10.2f %10.2f %10.2f\n", e.name, t[0] / base, t[1] / base, t[2] / base);
    }
    printf("1 unit = %.4f ns\n", base * 1e9);
    return 0;
}
