// asmc_flow_dev.h — device pieces of the coupling-flow log-density on the fp32 MFMA (v_mfma_f32_32x32x2_f32), shared by
// the stand-alone kernel (asmc_flow.hip) and the fused flow-proposal pCN step (asmc_pcn.hip).  Layout and packing are
// described at the top of asmc_flow.hip.
#pragma once
#include "asmc_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));


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
            const float v = b[(nb * 16 + r) * 2 + hh];
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
                                               const float* __restrict__ lp, int lane, int hh, float (&ladj)[TPW]) {
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
    for (int tt = 0; tt < TPW; tt++)
#pragma unroll
        for (int q = 0; q < H / 2; q++) {
            const float sraw = o[tt][q / 16][q % 16];
            const float t = o[tt][(H / 2 + q) / 16][(H / 2 + q) % 16];
            // s = 2 tanh(sraw / 2) = 2 - 4 / (exp(sraw) + 1) on the hardware exp2 / rcp units: absolute error
            // ~1e-7, which is all that matters (s is added to the log-determinant and exponentiated); libm's
            // tanhf + expf would cost as many issue cycles per layer as a third of its MFMAs
            const float s = 2.0f - 4.0f * __frcp_rn(__expf(sraw) + 1.0f);
            trans[tt][q] = (trans[tt][q] - t) * __expf(-s);
            ladj[tt] -= s;
        }
}

