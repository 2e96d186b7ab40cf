// asmc_flow_dev.h — device pieces of the coupling-flow log-density on the fp32 MFMA (v_mfma_f32_32x32x2_f32), shared by
// the stand-alone kernel (asmc_flow.hip) and the fused flow-proposal pCN step (asmc_pcn.hip).  Layout and packing are
// described at the top of asmc_flow.hip.
#pragma once
#include "asmc_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));


// the flow's standardisation (x - loc) / scale through the reciprocal y = rn(1 / scale): q0 = a y, r = a - scale q0 (exact in
// the FMA), q = q0 + r y - the correctly rounded quotient in three instructions instead of the ten of an IEEE division
__device__ __forceinline__ float flow_standardise(float x, float loc, float scale, float rcp) {
    const float a = x - loc;
    const float q0 = a * rcp;
    const float r = fmaf(-scale, q0, a);
    return fmaf(r, rcp, q0);
}

// two coordinates at once on the packed fp32 instructions (v_pk_add / mul / fma_f32): the same four IEEE operations per element, so the
// same bits as flow_standardise, in four instructions per PAIR
typedef float flow_std2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ flow_std2 flow_standardise2(flow_std2 x, flow_std2 loc, flow_std2 scale, flow_std2 rcp) {
    const flow_std2 a = x - loc;
    const flow_std2 q0 = a * rcp;
    const flow_std2 r = __builtin_elementwise_fma(-scale, q0, a);
    return __builtin_elementwise_fma(r, rcp, q0);
}

// (s_raw, t) of a transformed coordinate -> the coordinate and the log-determinant.
// form 0 - this repository's flows: s = 2 tanh(s_raw / 2) (= 2 - 4 / (exp(s_raw) + 1)), z = (x - t) exp(-s), ladj -= s;
// form 1 - ASMC_AFFINE_SOFTCLIP, zuko's MonotonicAffineTransform with slope 1e-3 (flows/torch/flows.py:156-168 builds zuko
//   flows; zuko is absent from the build image, so this follows its documented arithmetic and is UNVERIFIED against it):
//   ls = s_raw / (1 + |s_raw| / ln(1000)), z = x exp(ls) + t, ladj += ls.
// INVERSE: the sampling direction.  ladj collects the log-determinant of x -> z in both directions.
template <bool INVERSE>
__device__ __forceinline__ void flow_affine(float& x, float sraw, float t, float& ladj, int form) {
    if (form == 0) {
        const float sv = 2.0f - 4.0f * __builtin_amdgcn_rcpf(__expf(sraw) + 1.0f);
        if (INVERSE) {
            const float m = x * __expf(sv);
            x = m + t;
        } else {
            x = (x - t) * __expf(-sv);
        }
        ladj -= sv;
    } else {
        const float ls = sraw * __builtin_amdgcn_rcpf(1.0f + __builtin_fabsf(sraw) * 0.14476482730108395f);  // 1 / ln(1000)
        if (INVERSE) {
            x = (x - t) * __expf(-ls);
        } else {
            const float m = x * __expf(ls);
            x = m + t;
        }
        ladj += ls;
    }
}

// TWO coordinates of an affine transform at once, on the packed fp32 instructions (v_pk_mul / v_pk_add / v_pk_fma_f32: two results per
// lane and issue slot) - round 6: the scalar form cost 9.5 vector instructions per coordinate, this one 6.5 (three of them the
// transcendentals).  Form 0 per coordinate: e = exp2(s_raw log2 e), r = 1 / (e + 1), s = fma(r, -4, 2) [= 2 tanh(s_raw / 2)],
// z = (x - t) exp2(fma(r, 4 log2 e, -2 log2 e)) [= (x - t) exp(-s), the exponent formed from r by ONE fma instead of through s];
// sampling direction: x = fma(z, exp2(fma(r, -4 log2 e, 2 log2 e)), t).  `lacc` collects the layer's log-determinant per PARITY of the
// coordinate (the caller adds lacc.x + lacc.y to its running value at the end of the layer): every kernel family pairs and folds
// the same way (coupling_layer, _hs, _hs2, _hs1p, f16_layer), so a carried log q and its re-evaluation still agree bit for bit.
typedef float flow_f2 __attribute__((ext_vector_type(2)));
template <bool INVERSE>
__device__ __forceinline__ void flow_affine2(float& x0, float& x1, float s0, float s1, float t0, float t1, flow_f2& lacc, int form) {
    constexpr float L2E = 1.4426950408889634f;
    const flow_f2 sr = {s0, s1}, tt = {t0, t1};
    flow_f2 xx = {x0, x1};
    if (form == 0) {
        const flow_f2 a = sr * L2E;
        const flow_f2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        const flow_f2 d = e + 1.0f;
        const flow_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const flow_f2 sv = __builtin_elementwise_fma(r, (flow_f2)(-4.0f), (flow_f2)(2.0f));
        if (INVERSE) {
            const flow_f2 ea = __builtin_elementwise_fma(r, (flow_f2)(-4.0f * L2E), (flow_f2)(2.0f * L2E));
            const flow_f2 w = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
            xx = __builtin_elementwise_fma(xx, w, tt);
        } else {
            const flow_f2 ea = __builtin_elementwise_fma(r, (flow_f2)(4.0f * L2E), (flow_f2)(-2.0f * L2E));
            const flow_f2 w = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
            xx = (xx - tt) * w;
        }
        lacc -= sv;
    } else {
        const flow_f2 ab = {__builtin_fabsf(s0), __builtin_fabsf(s1)};
        const flow_f2 dn = __builtin_elementwise_fma(ab, (flow_f2)(0.14476482730108395f), (flow_f2)(1.0f));  // 1 + |s_raw| / ln(1000)
        const flow_f2 ls = sr * (flow_f2){__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
        if (INVERSE) {
            const flow_f2 ea = ls * (-L2E);
            xx = (xx - tt) * (flow_f2){__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
        } else {
            const flow_f2 ea = ls * L2E;
            xx = __builtin_elementwise_fma(xx, (flow_f2){__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)}, tt);
        }
        lacc += ls;
    }
    x0 = xx.x, x1 = xx.y;
}

template <int H, int W>
struct FlowDims {
    static constexpr int NB1 = W / 32;  // accumulator blocks of a hidden layer
    static constexpr int NB3 = H / 16;  // accumulator blocks of the output layer (2H rows)
    static constexpr int BIAS = (2 * NB1 + NB3) * 32;
    static constexpr int LAYER = BIAS + W * H + W * W + 2 * H * W;  // floats per coupling layer
};

__host__ __device__ static inline int acc_row(int r, int hh) { return 8 * (r / 4) + 4 * hh + (r % 4); }

// TPW = tiles (of 32 particles) per wave.  With two tiles a wave runs two independent accumulator chains that
// share every A operand: half the LDS reads per MFMA, and the vector work of one tile (ReLU, tanh/exp, hazards
// after a chain) issues in the shadow of the other tile's MFMAs instead of leaving the matrix pipe idle.
template <int TPW, int NB>
__device__ __forceinline__ void acc_bias(floatx16 (&acc)[TPW][NB], const float* __restrict__ b, int hh) {
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = b[hh * NB * 16 + nb * 16 + r];
#pragma unroll
            for (int tt = 0; tt < TPW; tt++) acc[tt][nb][r] = v;
        }
}

template <int TPW, int NB>
__device__ __forceinline__ void acc_relu(floatx16 (&acc)[TPW][NB]) {
#pragma unroll
    for (int tt = 0; tt < TPW; tt++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tt][nb][r] = fmaxf(acc[tt][nb][r], 0.0f);
}

// out[NBO] += Wt * in, `in` given as NBI accumulator blocks of the previous layer.  Consecutive MFMAs never hit the
// same accumulator: the output blocks (or, for a single output block, two partial accumulators over the even and
// odd k-groups) alternate, so the matrix pipe does not wait on the previous instruction's write-back.
template <int TPW, int NBO, int NBI>
__device__ __forceinline__ void dense_from_acc(floatx16 (&out)[TPW][NBO], const floatx16 (&in)[TPW][NBI],
                                               const float* __restrict__ A, int lane) {
    constexpr int G = NBI * 4;  // groups of four k-steps per output block
    const float4* Ap = reinterpret_cast<const float4*>(A) + lane;
    if constexpr (NBO >= 2) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            float av[NBO][4];
#pragma unroll
            for (int nbo = 0; nbo < NBO; nbo++) {
                const float4 a = Ap[(size_t)(nbo * G + g) * 64];
                av[nbo][0] = a.x, av[nbo][1] = a.y, av[nbo][2] = a.z, av[nbo][3] = a.w;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int s = 4 * g + e;
#pragma unroll
                for (int nbo = 0; nbo < NBO; nbo++)
#pragma unroll
                    for (int tt = 0; tt < TPW; tt++)
                        out[tt][nbo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nbo][e], in[tt][s / 16][s % 16], out[tt][nbo], 0, 0, 0);
            }
        }
    } else {
        floatx16 part[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) part[tt][r] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            const float4 a0 = Ap[(size_t)g * 64], a1 = Ap[(size_t)(g + 1) * 64];
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int s0 = 4 * g + e, s1 = 4 * (g + 1) + e;
#pragma unroll
                for (int tt = 0; tt < TPW; tt++) {
                    out[tt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], in[tt][s0 / 16][s0 % 16], out[tt][0], 0, 0, 0);
                    part[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], in[tt][s1 / 16][s1 % 16], part[tt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < TPW; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) out[tt][0][r] += part[tt][r];
    }
}

template <int H, int W, int TPW>
__device__ __forceinline__ void coupling_layer(const float (&cond)[TPW][H / 2], float (&trans)[TPW][H / 2],
                                               const float* __restrict__ lp, int lane, int hh, float (&ladj)[TPW], int form = 0) {
    using FD = FlowDims<H, W>;
    const float* b1 = lp;
    const float* b2 = b1 + FD::NB1 * 32;
    const float* b3 = b2 + FD::NB1 * 32;
    const float* A1 = b3 + FD::NB3 * 32;
    const float* A2 = A1 + W * H;
    const float* A3 = A2 + W * W;
    floatx16 h1[TPW][FD::NB1];
    acc_bias<TPW, FD::NB1>(h1, b1, hh);
    {
        constexpr int G1 = H / 8;
        const float4* Ap = reinterpret_cast<const float4*>(A1) + lane;
#pragma unroll
        for (int g = 0; g < G1; g++) {
            float av[FD::NB1][4];
#pragma unroll
            for (int nb = 0; nb < FD::NB1; nb++) {
                const float4 a = Ap[(size_t)(nb * G1 + g) * 64];
                av[nb][0] = a.x, av[nb][1] = a.y, av[nb][2] = a.z, av[nb][3] = a.w;
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int nb = 0; nb < FD::NB1; nb++)
#pragma unroll
                    for (int tt = 0; tt < TPW; tt++)
                        h1[tt][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nb][e], cond[tt][4 * g + e], h1[tt][nb], 0, 0, 0);
        }
    }
    acc_relu<TPW, FD::NB1>(h1);
    floatx16 h2[TPW][FD::NB1];
    acc_bias<TPW, FD::NB1>(h2, b2, hh);
    dense_from_acc<TPW, FD::NB1, FD::NB1>(h2, h1, A2, lane);
    acc_relu<TPW, FD::NB1>(h2);
    floatx16 o[TPW][FD::NB3];
    acc_bias<TPW, FD::NB3>(o, b3, hh);
    dense_from_acc<TPW, FD::NB3, FD::NB1>(o, h2, A3, lane);
#pragma unroll
    for (int tt = 0; tt < TPW; tt++) {
        static_assert((H / 2) % 2 == 0, "coordinates in pairs");
        flow_f2 lacc = {0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < H / 2; q += 2) {
            // s = 2 tanh(sraw / 2) = 2 - 4 / (exp(sraw) + 1) on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32): absolute error
            // ~1e-7, which is all that matters (s is added to the log-determinant and exponentiated); libm's
            // tanhf + expf would cost as many issue cycles per layer as a third of its MFMAs
            flow_affine2<false>(trans[tt][q], trans[tt][q + 1], o[tt][q / 16][q % 16], o[tt][(q + 1) / 16][(q + 1) % 16],
                                o[tt][(H / 2 + q) / 16][(H / 2 + q) % 16], o[tt][(H / 2 + q + 1) / 16][(H / 2 + q + 1) % 16], lacc, form);
        }
        ladj[tt] += lacc.x + lacc.y;
    }
}



// ---- split-fp16 form of the same layers (v_mfma_f32_32x32x16_f16) ----------------------------------------------------
// An fp32 value x is carried as the fp16 pair (hi, lo) = (rn16(x), rn16(x - hi)): x = hi + lo up to 2^-24 |x| (2^-25
// absolute below 2^-2; gfx950's matrix core honours fp16 subnormals on input - tools/ubench/mfma_f16_denorm.hip - so the
// residual keeps its bits there).  A layer's product is accumulated in fp32 as  Wh xh + Wh xl + Wl xh  (the dropped
// Wl xl is below 2^-24 of the term): three fp16 MFMAs of K = 16 do the work of eight fp32 MFMAs of K = 2 in 3/16 of the
// matrix-pipe time, at the accuracy of fp32 operands.  Range: |x| < 65504 for activations and weights (beyond that the
// pair is inf/NaN and the proposal is rejected like any other NaN density).
// The operand images are a regrouping of the fp32 pack: K16 step S of an output block takes the k-steps 8S .. 8S+7 of
// the fp32 layout, i.e. the float4 slots 2G' and 2G'+1 (G' = block * steps + S) of the (group, lane) order become the
// lane's eight hi halves and eight lo halves in those same two 16-byte slots (flow_stage_hs converts while staging the
// weights into LDS).  The B operand of step S is registers 8 (S % 2) .. + 7 of accumulator block S / 2 of the previous
// layer, converted lane-locally.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// `amax` collects the largest operand magnitude a lane has converted: past FLOW_HS_MAX the fp16 pair is inf / NaN and -
// ReLU squashing NaNs - the density would come out FINITE AND WRONG, so the callers turn it into NaN (rejected and counted)
#define FLOW_HS_MAX 65504.0f
template <bool RELU>
__device__ __forceinline__ void split8_f16(const float (&x)[8], half8& hi, half8& lo, float& amax) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float v0 = RELU ? __int_as_float(max(__float_as_int(x[j]), 0)) : x[j];
        const float v1 = RELU ? __int_as_float(max(__float_as_int(x[j + 1]), 0)) : x[j + 1];
        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v0)), __builtin_fabsf(v1));  // one v_max3_f32
    }
    // ReLU on the bit pattern: one v_max_i32 (negative floats are negative integers; fmaxf would canonicalise the MFMA
    // result first, two instructions)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = RELU ? __int_as_float(max(__float_as_int(x[j]), 0)) : x[j];
#ifdef FLOW_SPLIT_CVT
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const _Float16 h = (_Float16)v[j];
        hi[j] = h;
        lo[j] = (_Float16)(v[j] - (float)h);
    }
#else
    // lo = rn16(v - hi) as ONE mixed-precision FMA per element (fp16 hi taken from its half of the packed register, times
    // -1, plus the fp32 v; v - hi is exact in fp32, so the single rounding is the conversion's), written straight into
    // its half of the packed lo register: 3 instructions per pair instead of cvt_pk + 2 cvt + pk_add + cvt_pk.  hipcc
    // folds fma(h, -1, v) back into a subtraction, hence the asm; its inputs are vector-ALU results (v_max_i32 /
    // v_cvt_pk_f16_f32 / register moves, never an MFMA's D), and the trailing s_nop 1 covers the packed registers' use as
    // an MFMA operand (cdna_hip_programming.md 5.7 item 2).
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    half2v hp[4], lp[4];
#pragma unroll
    for (int j = 0; j < 4; j++) hp[j] = half2v{(_Float16)v[2 * j], (_Float16)v[2 * j + 1]};
    asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(lp[0]), "=&v"(lp[1]), "=&v"(lp[2]), "=&v"(lp[3])
        : "v"(hp[0]), "v"(hp[1]), "v"(hp[2]), "v"(hp[3]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]),
          "v"(v[6]), "v"(v[7]));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        hi[2 * j] = hp[j][0];
        hi[2 * j + 1] = hp[j][1];
        lo[2 * j] = lp[j][0];
        lo[2 * j + 1] = lp[j][1];
    }
#endif
}

// out[NBO] += Wt * act(in), `in` = NBI accumulator blocks of the previous layer (pre-activation), A = the matrix's image
template <int NBO, int NBI, bool RELU>
__device__ __forceinline__ void dense_from_acc_hs(floatx16 (&out)[NBO], const floatx16 (&in)[NBI], const float* __restrict__ A,
                                                  int lane, float& amax) {
    constexpr int ST = NBI * 2;  // K16 steps
    const half8* Ap = reinterpret_cast<const half8*>(A) + lane;
#pragma unroll
    for (int S = 0; S < ST; S++) {
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) xv[j] = in[S / 2][8 * (S % 2) + j];
        half8 bh, bl;
        split8_f16<RELU>(xv, bh, bl, amax);
        half8 ah[NBO], al[NBO];
#pragma unroll
        for (int nbo = 0; nbo < NBO; nbo++) {
            ah[nbo] = Ap[(size_t)(2 * (nbo * ST + S)) * 64];
            al[nbo] = Ap[(size_t)(2 * (nbo * ST + S) + 1) * 64];
        }
#pragma unroll
        for (int nbo = 0; nbo < NBO; nbo++) out[nbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[nbo], bh, out[nbo], 0, 0, 0);
#pragma unroll
        for (int nbo = 0; nbo < NBO; nbo++) out[nbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[nbo], bl, out[nbo], 0, 0, 0);
#pragma unroll
        for (int nbo = 0; nbo < NBO; nbo++) out[nbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[nbo], bh, out[nbo], 0, 0, 0);
#ifdef FLOW_HS4
#pragma unroll
        for (int nbo = 0; nbo < NBO; nbo++) out[nbo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[nbo], bl, out[nbo], 0, 0, 0);
#endif
    }
}

template <int NB>
__device__ __forceinline__ void acc_bias1(floatx16 (&acc)[NB], const float* __restrict__ b, int hh) {
    // contiguous per lane half, read as 16-byte vectors.  (Written as sixteen scalar reads, the two layer-parity branches of the fused
    // step shared their FIRST dword: the compiler hoisted that one read above the branch, fetched the other fifteen as misaligned
    // ds_read2_b32 pairs and reassembled the accumulator with 22 moves per layer - round 6, from the generated code.)
    typedef float bias_f4 __attribute__((ext_vector_type(4)));
    const bias_f4* __restrict__ bp = reinterpret_cast<const bias_f4*>(b + hh * NB * 16);
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bias_f4 t = bp[nb * 4 + q];
            acc[nb][4 * q] = t.x, acc[nb][4 * q + 1] = t.y, acc[nb][4 * q + 2] = t.z, acc[nb][4 * q + 3] = t.w;
        }
}

// one coupling layer of one 32-particle tile: cond / trans are the lane half's H / 2 coordinates
// INVERSE: the sampling direction, x_b = z_b exp(s) + t (flows.py _Coupling.inverse); ladj collects -s in both directions, so
// that base(z) + ladj is log q of the sample as well
template <int H, int W, bool INVERSE = false, int FORM = -1>  // (FORM: see coupling_layer_hs1p)
__device__ __forceinline__ void coupling_layer_hs(const float (&cond)[H / 2], float (&trans)[H / 2], const float* __restrict__ lp,
                                                  int lane, int hh, float& ladj, float& amax, int form = 0) {
    using FD = FlowDims<H, W>;
    const float* b1 = lp;
    const float* b2 = b1 + FD::NB1 * 32;
    const float* b3 = b2 + FD::NB1 * 32;
    const float* A1 = b3 + FD::NB3 * 32;
    const float* A2 = A1 + W * H;
    const float* A3 = A2 + W * W;
    floatx16 h1[FD::NB1];
    acc_bias1<FD::NB1>(h1, b1, hh);
    {
        constexpr int ST = H / 16;  // K16 steps over the conditioner inputs
        const half8* Ap = reinterpret_cast<const half8*>(A1) + lane;
#pragma unroll
        for (int S = 0; S < ST; S++) {
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) xv[j] = cond[8 * S + j];
            half8 bh, bl;
            split8_f16<false>(xv, bh, bl, amax);
            half8 ah[FD::NB1], al[FD::NB1];
#pragma unroll
            for (int nb = 0; nb < FD::NB1; nb++) {
                ah[nb] = Ap[(size_t)(2 * (nb * ST + S)) * 64];
                al[nb] = Ap[(size_t)(2 * (nb * ST + S) + 1) * 64];
            }
#pragma unroll
            for (int nb = 0; nb < FD::NB1; nb++) h1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[nb], bh, h1[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < FD::NB1; nb++) h1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[nb], bl, h1[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < FD::NB1; nb++) h1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[nb], bh, h1[nb], 0, 0, 0);
#ifdef FLOW_HS4
#pragma unroll
            for (int nb = 0; nb < FD::NB1; nb++) h1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[nb], bl, h1[nb], 0, 0, 0);
#endif
        }
    }
    floatx16 h2[FD::NB1];
    acc_bias1<FD::NB1>(h2, b2, hh);
    dense_from_acc_hs<FD::NB1, FD::NB1, true>(h2, h1, A2, lane, amax);
    floatx16 o[FD::NB3];
    acc_bias1<FD::NB3>(o, b3, hh);
    dense_from_acc_hs<FD::NB3, FD::NB1, true>(o, h2, A3, lane, amax);
    flow_f2 lacc = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < H / 2; q += 2)  // (form 0: s = 2 tanh(sraw / 2), see coupling_layer; pairs: flow_affine2)
        flow_affine2<INVERSE>(trans[q], trans[q + 1], o[q / 16][q % 16], o[(q + 1) / 16][(q + 1) % 16], o[(H / 2 + q) / 16][(H / 2 + q) % 16],
                              o[(H / 2 + q + 1) / 16][(H / 2 + q + 1) % 16], lacc, FORM >= 0 ? FORM : form);
    ladj += lacc.x + lacc.y;
}

// ---- two tiles through one coupling layer, matrix and vector work interleaved by hand (round 3) ---------------------------
// coupling_layer_hs leaves the order of a tile's instructions to the compiler, which emits a K-step's conversions (16-24
// vector instructions) and then its MFMAs back to back (3-6 of them, 32 cycles each): while a burst runs the wave has no
// vector instruction to issue, and while it converts the matrix pipe idles - in the fused flow step (two waves per SIMD)
// the two pipes ran one after the other: vector ALU active 58 %, matrix pipe 24 %, sum 83 % of the kernel.  Here the wave's TWO
// flow tiles (A, B) go through the layer together, one K-step group at a time, and every MFMA of one tile is followed by a
// quarter of the OTHER tile's next operand conversion (two elements: ReLU, range check, hi pair, lo pair - about one MFMA's
// worth of issue cycles), fenced by scheduling barriers so that the order survives:
//     cvt A(0);  for g:  { MFMAs A(g) | cvt B(g) }  { MFMAs B(g) | cvt A(g + 1) }  ...  { MFMAs B(last) | epilogue A };  epilogue B
// The A operands of a group are read once for both tiles.  Same operations in the same order per accumulator as
// coupling_layer_hs: bit-identical results.
typedef unsigned flow_u4 __attribute__((ext_vector_type(4)));

// elements (x0, x1) -> their packed hi halves and lo halves (see split8_f16).  The range check rides on the hi halves
// (split4_range below): an element past the fp16 range converts to inf, and 65504 itself is flagged like before.
// NOP: end the statement with the two wait states a just-written VGPR needs before an MFMA reads it as an operand
// (cdna_hip_programming.md 5.7 item 2) - only the LAST quarter of an operand can be followed directly by its consumer.
// PROP (round 6, the one-kernel step kernels): the ReLU is the IEEE-754-2019 maximum (v_maximum3_f32 x, 0, 0 - one instruction, like the
// integer maximum), which PROPAGATES a NaN instead of squashing one of negative sign to zero.  An element past the fp16 range then needs
// no range check at all: it converts to (hi, lo) = (+-inf, -+inf), the products with any weight - zero included - accumulate to NaN, and
// the NaN reaches log q, where the step rejects and counts it exactly as the explicit check's NaN was.  The stand-alone density / sampling
// kernels keep the explicit check (a sampling pass must return finite positions): same bits wherever nothing overflowed.
template <bool RELU, bool NOP = true, bool PROP = false>
__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned& hp, unsigned& lp) {
    const float v0 = !RELU ? x0 : PROP ? __builtin_elementwise_maximum(x0, 0.0f) : __int_as_float(max(__float_as_int(x0), 0));
    const float v1 = !RELU ? x1 : PROP ? __builtin_elementwise_maximum(x1, 0.0f) : __int_as_float(max(__float_as_int(x1), 0));
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v h = half2v{(_Float16)v0, (_Float16)v1};
    half2v l;
    if (NOP)
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "s_nop 1"
            : "=&v"(l)
            : "v"(h), "v"(v0), "v"(v1));
    else
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(l)
            : "v"(h), "v"(v0), "v"(v1));
    hp = __builtin_bit_cast(unsigned, h);
    lp = __builtin_bit_cast(unsigned, l);
}
// largest magnitude among the hi halves of FOUR elements (two packed registers) folded into the packed running maximum:
// one v_pk_maximum3_f16 (gfx950; IEEE maximum, so a NaN sticks) - after a ReLU the halves are non-negative; signed inputs
// take a second instruction on the negated pairs.  amax_pk starts at 0.
template <bool SIGNED>
__device__ __forceinline__ void split4_range(unsigned h0, unsigned h1, unsigned& amax_pk) {
    asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(amax_pk) : "v"(h0), "v"(h1));
    if (SIGNED) asm("v_pk_maximum3_f16 %0, %0, %1, %2 neg_lo:[0,1,1] neg_hi:[0,1,1]" : "+v"(amax_pk) : "v"(h0), "v"(h1));
}
// the two halves of the packed running maximum -> the fp32 maximum the callers compare with FLOW_HS_MAX
__device__ __forceinline__ float range_pk_max(unsigned amax_pk) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v a = __builtin_bit_cast(half2v, amax_pk);
    const float a0 = (float)a[0], a1 = (float)a[1];
    return (a0 != a0 || a1 != a1) ? __builtin_nanf("") : __builtin_fmaxf(a0, a1);
}

template <int H, int W>
__device__ __forceinline__ void coupling_layer_hs2(const float (&condA)[H / 2], float (&transA)[H / 2], const float (&condB)[H / 2],
                                                   float (&transB)[H / 2], const float* __restrict__ lp, int lane, int hh,
                                                   float& ladjA, float& ladjB, unsigned& amaxA, unsigned& amaxB) {
    using FD = FlowDims<H, W>;
    static_assert(H == 16, "one K16 step over the conditioner inputs");
    constexpr int NB1 = FD::NB1, NB3 = FD::NB3, ST1 = H / 16, ST2 = 2 * NB1, G = ST1 + 2 * ST2;
    const float* b1 = lp;
    const float* b2 = b1 + NB1 * 32;
    const float* b3 = b2 + NB1 * 32;
    const float* a1f = b3 + NB3 * 32;
    const half8* A1 = reinterpret_cast<const half8*>(a1f) + lane;
    const half8* A2 = reinterpret_cast<const half8*>(a1f + W * H) + lane;
    const half8* A3 = reinterpret_cast<const half8*>(a1f + W * H + W * W) + lane;
    floatx16 h1A[NB1], h1B[NB1], h2A[NB1], h2B[NB1], oA[NB3], oB[NB3];
    acc_bias1<NB1>(h1A, b1, hh);  // (each tile reads its own copy of the bias: LDS reads cost no vector instruction, moves would)
    acc_bias1<NB1>(h1B, b1, hh);
    // source element j of group g's operand: the conditioner input (g < ST1), then h1 (pre-activation), then h2
    auto src = [&](const float (&cond)[H / 2], const floatx16 (&h1)[NB1], const floatx16 (&h2)[NB1], int g, int j) -> float {
        if (g < ST1) return cond[8 * g + j];
        const int S = (g - ST1) % ST2;
        return g < ST1 + ST2 ? h1[S / 2][8 * (S % 2) + j] : h2[S / 2][8 * (S % 2) + j];
    };
    unsigned hpA[4], lpA[4], hpB[4], lpB[4];
    auto cvtA = [&](int g, int c) {  // quarter c of group g's operand; the odd quarters also range-check their pair of quarters
        const float e0 = src(condA, h1A, h2A, g, 2 * c), e1 = src(condA, h1A, h2A, g, 2 * c + 1);
        if (c == 3) {  // the quarter that may be followed directly by the MFMAs that read the operand: wait states inside
            if (g < ST1) split2_f16<false, true>(e0, e1, hpA[c], lpA[c]);
            else split2_f16<true, true>(e0, e1, hpA[c], lpA[c]);
        } else {
            if (g < ST1) split2_f16<false, false>(e0, e1, hpA[c], lpA[c]);
            else split2_f16<true, false>(e0, e1, hpA[c], lpA[c]);
        }
        if (c & 1) {
            if (g < ST1) split4_range<true>(hpA[c - 1], hpA[c], amaxA);
            else split4_range<false>(hpA[c - 1], hpA[c], amaxA);
        }
    };
    auto cvtB = [&](int g, int c) {
        const float e0 = src(condB, h1B, h2B, g, 2 * c), e1 = src(condB, h1B, h2B, g, 2 * c + 1);
        if (c == 3) {  // the quarter that may be followed directly by the MFMAs that read the operand: wait states inside
            if (g < ST1) split2_f16<false, true>(e0, e1, hpB[c], lpB[c]);
            else split2_f16<true, true>(e0, e1, hpB[c], lpB[c]);
        } else {
            if (g < ST1) split2_f16<false, false>(e0, e1, hpB[c], lpB[c]);
            else split2_f16<true, false>(e0, e1, hpB[c], lpB[c]);
        }
        if (c & 1) {
            if (g < ST1) split4_range<true>(hpB[c - 1], hpB[c], amaxB);
            else split4_range<false>(hpB[c - 1], hpB[c], amaxB);
        }
    };
    auto pack = [](const unsigned (&q)[4]) -> half8 { return __builtin_bit_cast(half8, flow_u4{q[0], q[1], q[2], q[3]}); };
    // the epilogue of one tile in four parts (two of its H / 2 coordinates each)
    flow_f2 laccA = {0.0f, 0.0f}, laccB = {0.0f, 0.0f};  // (folded into ladjA / ladjB behind the last part)
    auto epi = [&](const floatx16 (&o)[NB3], float (&trans)[H / 2], flow_f2& lacc, int c) {
        const int q = 2 * c;  // 2 tanh(sraw / 2), see coupling_layer; the pair (q, q + 1): flow_affine2
        flow_affine2<false>(trans[q], trans[q + 1], o[q / 16][q % 16], o[(q + 1) / 16][(q + 1) % 16], o[(H / 2 + q) / 16][(H / 2 + q) % 16],
                            o[(H / 2 + q + 1) / 16][(H / 2 + q + 1) % 16], lacc, 0);
    };
#pragma unroll
    for (int c = 0; c < 4; c++) cvtA(0, c);
    half8 bhA = pack(hpA), blA = pack(lpA), bhB, blB;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; g++) {
        // group g: layer, K-step, output blocks, operand image
        const int layer = g < ST1 ? 0 : g < ST1 + ST2 ? 1 : 2;
        const int S = layer == 0 ? g : (g - ST1) % ST2, ST = layer == 0 ? ST1 : ST2;
        const int NBO = layer == 2 ? NB3 : NB1;
        const half8* Ap = layer == 0 ? A1 : layer == 1 ? A2 : A3;
        half8 ah[NB1 > NB3 ? NB1 : NB3], al[NB1 > NB3 ? NB1 : NB3];
#pragma unroll
        for (int nb = 0; nb < NBO; nb++) {
            ah[nb] = Ap[(size_t)(2 * (nb * ST + S)) * 64];
            al[nb] = Ap[(size_t)(2 * (nb * ST + S) + 1) * 64];
        }
        if (layer == 1 && S == 0) {  // the accumulators of the layer that starts here: bias
            acc_bias1<NB1>(h2A, b2, hh);
            acc_bias1<NB1>(h2B, b2, hh);
        }
        if (layer == 2 && S == 0) {
            acc_bias1<NB3>(oA, b3, hh);
            acc_bias1<NB3>(oB, b3, hh);
        }
        const int NM = 3 * NBO;
        // { MFMAs A(g) | cvt B(g) }
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const int k = m / NBO, nb = m % NBO;
            floatx16& acc = layer == 0 ? h1A[nb] : layer == 1 ? h2A[nb] : oA[nb];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? ah[nb] : al[nb], k == 1 ? blA : bhA, acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (c * NM / 4 == m) cvtB(g, c);
            __builtin_amdgcn_sched_barrier(0);
        }
        bhB = pack(hpB), blB = pack(lpB);
        // { MFMAs B(g) | cvt A(g + 1) }, the last group: { MFMAs B | epilogue A } - whose o is complete since the slot before
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const int k = m / NBO, nb = m % NBO;
            floatx16& acc = layer == 0 ? h1B[nb] : layer == 1 ? h2B[nb] : oB[nb];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? ah[nb] : al[nb], k == 1 ? blB : bhB, acc, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (c * NM / 4 == m) {
                    if (g + 1 < G) cvtA(g + 1, c);
                    else epi(oA, transA, laccA, c);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g + 1 < G) bhA = pack(hpA), blA = pack(lpA);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) epi(oB, transB, laccB, c);
    ladjA += laccA.x + laccA.y;
    ladjB += laccB.x + laccB.y;
}

// ---- ONE tile through a coupling layer, its own conversions in the shadow of its own MFMAs (round 4) --------------------------
// coupling_layer_hs2 hides a tile's operand conversions behind the OTHER tile's MFMAs, which keeps two tiles' accumulators alive
// (128 registers at W = 64) and leaves no room for anything else across the flow.  Here ONE tile runs with its conversions
// software-pipelined against its own MFMAs: while the MFMAs of K-step group g issue, the operand of group g + 1 is converted a
// quarter at a time - possible whenever group g + 1 reads the PREVIOUS layer's accumulators (complete since that layer ended).
// Only the first group of a layer cannot start early (its source is what group g is still accumulating): three short bubbles per
// coupling layer, which the SIMD's other wave fills.  Peak accumulators: 64 registers - the fused pCN step keeps the proposal y'
// (64 registers) in the register file across the flow instead of parking it in HBM.  PREFETCH: the A operands of group g + 1 are
// read from LDS while group g's MFMAs issue (16 more registers at W = 64).
// Same operations in the same order per accumulator as coupling_layer_hs / _hs2: bit-identical results.
// FORM: the affine form (flow_affine) as a compile-time constant, or -1 for the run-time argument (then every coordinate of the epilogue
// carries a uniform branch).
template <int H, int W, bool PREFETCH = true, int FORM = -1, bool PROP = false>  // (PROP: no range check, NaN-propagating ReLU - split2_f16)
__device__ __forceinline__ void coupling_layer_hs1p(const float (&cond)[H / 2], float (&trans)[H / 2], const float* __restrict__ lp,
                                                    int lane, int hh, float& ladj, unsigned& amax, int form = 0) {
    using FD = FlowDims<H, W>;
    constexpr int NB1 = FD::NB1, NB3 = FD::NB3, ST1 = H / 16, ST2 = 2 * NB1, G = ST1 + 2 * ST2;
    constexpr int NBM = NB1 > NB3 ? NB1 : NB3;
    const float* b1 = lp;
    const float* b2 = b1 + NB1 * 32;
    const float* b3 = b2 + NB1 * 32;
    const float* a1f = b3 + NB3 * 32;
    const half8* A1 = reinterpret_cast<const half8*>(a1f) + lane;
    const half8* A2 = reinterpret_cast<const half8*>(a1f + W * H) + lane;
    const half8* A3 = reinterpret_cast<const half8*>(a1f + W * H + W * W) + lane;
#ifndef HS1P_EARLY_BIAS
#define HS1P_EARLY_BIAS 0  // 1: the second layer's accumulators take their bias before the first layer's MFMAs; 2: ... and the output layer's at the start of the second
#endif
    floatx16 h1[NB1], h2[NB1], o[NB3];
    acc_bias1<NB1>(h1, b1, hh);
    if (HS1P_EARLY_BIAS >= 1) acc_bias1<NB1>(h2, b2, hh);
    auto src = [&](int g, int j) -> float {
        if (g < ST1) return cond[8 * g + j];
        const int S = (g - ST1) % ST2;
        return g < ST1 + ST2 ? h1[S / 2][8 * (S % 2) + j] : h2[S / 2][8 * (S % 2) + j];
    };
    auto first_of_layer = [](int g) { return g == 0 || g == ST1 || g == ST1 + ST2; };
    unsigned hp[4], lq[4];
    auto cvt = [&](int g, int c, bool last) {  // quarter c of group g's operand; `last`: the MFMAs that read it may follow directly
        const float e0 = src(g, 2 * c), e1 = src(g, 2 * c + 1);
        if (last) {
            if (g < ST1) split2_f16<false, true, PROP>(e0, e1, hp[c], lq[c]);
            else split2_f16<true, true, PROP>(e0, e1, hp[c], lq[c]);
        } else {
            if (g < ST1) split2_f16<false, false, PROP>(e0, e1, hp[c], lq[c]);
            else split2_f16<true, false, PROP>(e0, e1, hp[c], lq[c]);
        }
        if (!PROP && (c & 1)) {
            if (g < ST1) split4_range<true>(hp[c - 1], hp[c], amax);
            else split4_range<false>(hp[c - 1], hp[c], amax);
        }
    };
    auto pack = [](const unsigned (&q)[4]) -> half8 { return __builtin_bit_cast(half8, flow_u4{q[0], q[1], q[2], q[3]}); };
    auto a_ptr = [&](int g) -> const half8* { return g < ST1 ? A1 : g < ST1 + ST2 ? A2 : A3; };
    auto a_load = [&](int g, half8 (&ah)[NBM], half8 (&al)[NBM]) {
        const int layer = g < ST1 ? 0 : g < ST1 + ST2 ? 1 : 2;
        const int S = layer == 0 ? g : (g - ST1) % ST2, ST = layer == 0 ? ST1 : ST2, NBO = layer == 2 ? NB3 : NB1;
        const half8* Ap = a_ptr(g);
#pragma unroll
        for (int nb = 0; nb < NBO; nb++) {
            ah[nb] = Ap[(size_t)(2 * (nb * ST + S)) * 64];
            al[nb] = Ap[(size_t)(2 * (nb * ST + S) + 1) * 64];
        }
    };
#pragma unroll
    for (int c = 0; c < 4; c++) cvt(0, c, c == 3);
    half8 bh = pack(hp), bl = pack(lq);
    half8 ah[NBM], al[NBM], ahn[NBM], aln[NBM];
    a_load(0, ah, al);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int layer = g < ST1 ? 0 : g < ST1 + ST2 ? 1 : 2;
        const int NBO = layer == 2 ? NB3 : NB1, NM = 3 * NBO;
        const bool more = g + 1 < G, overlap = more && !first_of_layer(g + 1);
        if (g == ST1 && HS1P_EARLY_BIAS < 1) acc_bias1<NB1>(h2, b2, hh);
        if (g == (HS1P_EARLY_BIAS >= 2 ? ST1 : ST1 + ST2)) acc_bias1<NB3>(o, b3, hh);
        if (PREFETCH && more) a_load(g + 1, ahn, aln);  // lands while this group's MFMAs issue
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const int k = m / NBO, nb = m % NBO;
            floatx16& acc = layer == 0 ? h1[nb] : layer == 1 ? h2[nb] : o[nb];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k < 2 ? ah[nb] : al[nb], k == 1 ? bl : bh, acc, 0, 0, 0);
            if (overlap) {
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (c * NM / 4 == m) cvt(g + 1, c, c == 3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            if (!overlap) {  // the next layer's first operand: its source is complete only now
#pragma unroll
                for (int c = 0; c < 4; c++) cvt(g + 1, c, c == 3);
            }
            bh = pack(hp), bl = pack(lq);
            if (PREFETCH) {
#pragma unroll
                for (int nb = 0; nb < NBM; nb++) ah[nb] = ahn[nb], al[nb] = aln[nb];
            } else {
                a_load(g + 1, ah, al);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    flow_f2 lacc = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < H / 2; q += 2)  // (form 0: 2 tanh(sraw / 2), see coupling_layer; pairs: flow_affine2)
        flow_affine2<false>(trans[q], trans[q + 1], o[q / 16][q % 16], o[(q + 1) / 16][(q + 1) % 16], o[(H / 2 + q) / 16][(H / 2 + q) % 16],
                            o[(H / 2 + q + 1) / 16][(H / 2 + q + 1) % 16], lacc, FORM >= 0 ? FORM : form);
    ladj += lacc.x + lacc.y;
}

// Stage `n_layers` coupling layers from the fp32 pack in HBM into LDS as split-fp16 operand images (biases copied).
template <int H, int W, int THREADS>
__device__ __forceinline__ void flow_stage_hs(float* __restrict__ sp, const float* __restrict__ packed, int n_layers) {
    using FD = FlowDims<H, W>;
    constexpr int L4 = FD::LAYER / 4, B4 = FD::BIAS / 4;
    const int total4 = n_layers * L4;
    for (int i4 = threadIdx.x; i4 < total4; i4 += THREADS) {
        const int in_layer = i4 % L4;
        if (in_layer < B4) {
            reinterpret_cast<float4*>(sp)[i4] = reinterpret_cast<const float4*>(packed)[i4];
            continue;
        }
        const int local = in_layer - B4;
        if ((local / 64) & 1) continue;  // the even slot group's thread converts the pair
        const float4 f0 = reinterpret_cast<const float4*>(packed)[i4], f1 = reinterpret_cast<const float4*>(packed)[i4 + 64];
        const float xv[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        half8 hi, lo;
        float wmax = 0.0f;  // (weights beyond the fp16 range are refused when the flow is packed: asmc_coupling_pack)
        split8_f16<false>(xv, hi, lo, wmax);
        reinterpret_cast<half8*>(sp)[i4] = hi;
        reinterpret_cast<half8*>(sp)[i4 + 64] = lo;
    }
}
