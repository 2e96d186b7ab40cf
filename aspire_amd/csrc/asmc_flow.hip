// asmc_flow.hip — log-density of an affine coupling flow (RealNVP) on the CDNA4 matrix cores.
//
// Replaces the torch round trip of Flow.log_prob (reference src/aspire/flows/torch/flows.py:368-387) inside
// the tempered log-target of every MCMC step (src/aspire/samplers/smc/base.py:507-519).  The reference's
// flow arithmetic lives in zuko (absent here, parity unpinned); the flow evaluated is this repository's
// CouplingFlow (aspire_amd/flows.py): standardise, then n_layers coupling layers with alternating
// first-half / second-half masks; conditioner MLP  d/2 -> W -> W -> d  (ReLU);  s = 2 tanh(s_raw / 2);
// z_b = (x_b - t) exp(-s);  log q(x) = N(z; 0, I) - sum s - sum log scale.   All arithmetic is fp32 (the flow's
// dtype; the reference's flows are fp32 as well, flows/torch/flows.py:31-35), the result is widened to fp64.
//
// Mapping.  This is the one GEMM-shaped piece of the path (57 kflop per particle at d = 32, W = 64), so
// it runs on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, bitwise an fmaf chain).  One wave owns a tile
// of 32 particles; the MFMA computes  D[unit][particle] += Wt[unit][k] * act[k][particle],  i.e. the weights
// are the A operand and the activations the B operand.  In the 32x32 accumulator layout lane l holds column
// (particle) l % 32 and the sixteen rows  u(r, l/32) = 8 (r/4) + 4 (l/32) + r % 4:  each lane half owns
// half of the hidden units of its particle.  The next layer contracts over exactly those units, and a
// k-step of the 32x32x2 instruction takes k = 0 from lanes 0-31 and k = 1 from lanes 32-63, so accumulator
// register r of one layer IS the B operand of k-step r of the next layer — the contraction order is merely
// permuted, and the permutation is folded into the weight packing on the host.  Activations therefore never
// leave the register file between layers; the only LDS traffic is the A operand (one ds_read_b128 per four
// MFMAs, conflict-free because the pack is stored in (mfma, lane) order).
// The two lane halves also split the particle's coordinates: half hh holds conditioner dims
// [hh H/2, (hh+1) H/2) and the same range of the transformed half, and the output layer's rows are packed so
// that (s_j, t_j) land in the lane that holds x_j.
//
// Weights stay resident in LDS when the whole flow fits (4 layers of d = 32, W = 64: 115 KB of the CU's
// 160 KB); otherwise each coupling layer's block is streamed in turn while the tile state waits in registers.
#include "asmc_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define FLOW_THREADS 512  // 8 waves = 2 per SIMD: one wave's MFMAs cover the other's LDS reads and tanh/exp
#define FLOW_WAVES (FLOW_THREADS / 64)

template <int H, int W>
struct FlowDims {
    static constexpr int NB1 = W / 32;  // accumulator blocks of a hidden layer
    static constexpr int NB3 = H / 16;  // accumulator blocks of the output layer (2H rows)
    static constexpr int BIAS = (2 * NB1 + NB3) * 32;
    static constexpr int LAYER = BIAS + W * H + W * W + 2 * H * W;  // floats per coupling layer
};

__host__ __device__ static inline int acc_row(int r, int hh) { return 8 * (r / 4) + 4 * hh + (r % 4); }

template <int NB>
__device__ __forceinline__ void acc_bias(floatx16 (&acc)[NB], const float* __restrict__ b, int hh) {
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nb][r] = b[(nb * 16 + r) * 2 + hh];
}

template <int NB>
__device__ __forceinline__ void acc_relu(floatx16 (&acc)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nb][r] = fmaxf(acc[nb][r], 0.0f);
}

// out[NBO] += Wt * in, `in` given as NBI accumulator blocks of the previous layer
template <int NBO, int NBI>
__device__ __forceinline__ void dense_from_acc(floatx16 (&out)[NBO], const floatx16 (&in)[NBI],
                                               const float* __restrict__ A, int lane) {
    constexpr int G = NBI * 4;  // groups of four k-steps
#pragma unroll
    for (int nbo = 0; nbo < NBO; nbo++) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float4 a = *reinterpret_cast<const float4*>(A + ((size_t)(nbo * G + g) * 64 + lane) * 4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int s = 4 * g + e;
                out[nbo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], in[s / 16][s % 16], out[nbo], 0, 0, 0);
            }
        }
    }
}

template <int H, int W>
__device__ __forceinline__ void coupling_layer(const float (&cond)[H / 2], float (&trans)[H / 2],
                                               const float* __restrict__ lp, int lane, int hh, float& ladj) {
    using FD = FlowDims<H, W>;
    const float* b1 = lp;
    const float* b2 = b1 + FD::NB1 * 32;
    const float* b3 = b2 + FD::NB1 * 32;
    const float* A1 = b3 + FD::NB3 * 32;
    const float* A2 = A1 + W * H;
    const float* A3 = A2 + W * W;
    floatx16 h1[FD::NB1];
    acc_bias<FD::NB1>(h1, b1, hh);
#pragma unroll
    for (int nb = 0; nb < FD::NB1; nb++) {
#pragma unroll
        for (int g = 0; g < H / 8; g++) {
            const float4 a = *reinterpret_cast<const float4*>(A1 + ((size_t)(nb * (H / 8) + g) * 64 + lane) * 4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                h1[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], cond[4 * g + e], h1[nb], 0, 0, 0);
        }
    }
    acc_relu<FD::NB1>(h1);
    floatx16 h2[FD::NB1];
    acc_bias<FD::NB1>(h2, b2, hh);
    dense_from_acc<FD::NB1, FD::NB1>(h2, h1, A2, lane);
    acc_relu<FD::NB1>(h2);
    floatx16 o[FD::NB3];
    acc_bias<FD::NB3>(o, b3, hh);
    dense_from_acc<FD::NB3, FD::NB1>(o, h2, A3, lane);
#pragma unroll
    for (int q = 0; q < H / 2; q++) {
        const float sraw = o[q / 16][q % 16];
        const float t = o[(H / 2 + q) / 16][(H / 2 + q) % 16];
        const float s = 2.0f * tanhf(sraw * 0.5f);
        trans[q] = (trans[q] - t) * expf(-s);
        ladj -= s;
    }
}

template <int H, int W, typename XT>
__global__ __launch_bounds__(FLOW_THREADS) void k_coupling_logprob(int64_t n, int d, const XT* __restrict__ x,
                                                                  const float* __restrict__ packed, int n_layers,
                                                                  int resident, const float* __restrict__ loc,
                                                                  const float* __restrict__ scale, float ladj0,
                                                                  float base_const, double* __restrict__ out) {
    extern __shared__ __align__(16) float sp[];
    using FD = FlowDims<H, W>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, hh = lane >> 5;
    const int dh = d / 2;
    if (resident) {
        for (int i = threadIdx.x * 4; i < n_layers * FD::LAYER; i += FLOW_THREADS * 4)
            *reinterpret_cast<float4*>(sp + i) = *reinterpret_cast<const float4*>(packed + i);
        __syncthreads();
    }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t tiles_per_round = (int64_t)gridDim.x * FLOW_WAVES;
    const int64_t rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;  // same trip count for every wave
    for (int64_t it = 0; it < rounds; it++) {
        const int64_t tile = (it * gridDim.x + blockIdx.x) * FLOW_WAVES + wave;
        const int64_t row = tile * 32 + p;
        const bool valid = tile < n_tiles && row < n;
        float xa[H / 2], xb[H / 2];
#pragma unroll
        for (int i = 0; i < H / 2; i++) {
            const int jp = hh * (H / 2) + i;
            xa[i] = 0.0f;
            xb[i] = 0.0f;
            if (valid && jp < dh) {
                xa[i] = ((float)x[row * d + jp] - loc[jp]) / scale[jp];
                xb[i] = ((float)x[row * d + dh + jp] - loc[dh + jp]) / scale[dh + jp];
            }
        }
        float ladj = 0.0f;
        for (int c = 0; c < n_layers; c++) {
            const float* lp = sp + (resident ? (size_t)c * FD::LAYER : 0);
            if (!resident) {
                __syncthreads();  // every wave is done with the previous layer's block
                for (int i = threadIdx.x * 4; i < FD::LAYER; i += FLOW_THREADS * 4)
                    *reinterpret_cast<float4*>(sp + i) =
                        *reinterpret_cast<const float4*>(packed + (size_t)c * FD::LAYER + i);
                __syncthreads();
            }
            if ((c & 1) == 0)
                coupling_layer<H, W>(xa, xb, lp, lane, hh, ladj);
            else
                coupling_layer<H, W>(xb, xa, lp, lane, hh, ladj);
        }
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < H / 2; i++) q += xa[i] * xa[i] + xb[i] * xb[i];
        q += __shfl_xor(q, 32);
        ladj += __shfl_xor(ladj, 32);
        if (valid && hh == 0) out[row] = (double)((-0.5f * q + base_const) + (ladj0 + ladj));
    }
}

// ---------------------------------------------------------------------------------------------
// host side
static int flow_half_pad(int dims) { return ((dims / 2 + 15) / 16) * 16; }

static bool flow_supported(int dims, int hidden) {
    const int H = flow_half_pad(dims);
    return dims >= 2 && dims % 2 == 0 && (H == 16 || H == 32) && (hidden == 32 || hidden == 64 || hidden == 128);
}

static int64_t flow_layer_floats(int H, int Wd) { return (int64_t)(2 * (Wd / 32) + H / 16) * 32 + (int64_t)Wd * H + (int64_t)Wd * Wd + 2LL * H * Wd; }

extern "C" int64_t asmc_coupling_pack_floats(int dims, int n_layers, int hidden) {
    if (!flow_supported(dims, hidden) || n_layers < 1) {
        asmc_set_error("asmc_coupling_pack_floats: unsupported flow (dims even and <= 64, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    return (int64_t)n_layers * flow_layer_floats(flow_half_pad(dims), hidden);
}

extern "C" int asmc_coupling_pack(int dims, int n_layers, int hidden, const float* const* weights_host,
                                  const float* const* biases_host, float* packed_host) {
    ASMC_REQUIRE(weights_host && biases_host && packed_host, "null pointer");
    if (!flow_supported(dims, hidden) || n_layers < 1) {
        asmc_set_error("asmc_coupling_pack: unsupported flow (dims even and <= 64, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    const int H = flow_half_pad(dims), Wd = hidden, dh = dims / 2;
    const int NB1 = Wd / 32, NB3 = H / 16;
    const int64_t layer = flow_layer_floats(H, Wd);
    auto unit = [](int s, int hh) { return 32 * (s / 16) + acc_row(s % 16, hh); };  // s-th hidden unit held by half hh
    // packed output row (block nb, row i) -> row of the torch output layer ([s_0..s_dh-1, t_0..t_dh-1]) or -1
    auto out_row = [&](int nb, int i) {
        const int hh = (i / 4) % 2, r = 4 * (i / 8) + i % 4, q = 16 * nb + r;
        const int jp = hh * (H / 2) + (q < H / 2 ? q : q - H / 2);
        if (jp >= dh) return -1;
        return q < H / 2 ? jp : dh + jp;
    };
    for (int c = 0; c < n_layers; c++) {
        const float *W1 = weights_host[3 * c], *W2 = weights_host[3 * c + 1], *W3 = weights_host[3 * c + 2];
        const float *B1 = biases_host[3 * c], *B2 = biases_host[3 * c + 1], *B3 = biases_host[3 * c + 2];
        float* b1 = packed_host + c * layer;
        float* b2 = b1 + NB1 * 32;
        float* b3 = b2 + NB1 * 32;
        float* A1 = b3 + NB3 * 32;
        float* A2 = A1 + (int64_t)Wd * H;
        float* A3 = A2 + (int64_t)Wd * Wd;
        for (int nb = 0; nb < NB1; nb++)
            for (int r = 0; r < 16; r++)
                for (int hh = 0; hh < 2; hh++) {
                    b1[(nb * 16 + r) * 2 + hh] = B1[32 * nb + acc_row(r, hh)];
                    b2[(nb * 16 + r) * 2 + hh] = B2[32 * nb + acc_row(r, hh)];
                }
        for (int nb = 0; nb < NB3; nb++)
            for (int r = 0; r < 16; r++)
                for (int hh = 0; hh < 2; hh++) {
                    const int o = out_row(nb, acc_row(r, hh));
                    b3[(nb * 16 + r) * 2 + hh] = o >= 0 ? B3[o] : 0.0f;
                }
        for (int nb = 0; nb < NB1; nb++)
            for (int g = 0; g < H / 8; g++)
                for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 4; e++) {
                        const int in = (l / 32) * (H / 2) + 4 * g + e;
                        A1[((int64_t)(nb * (H / 8) + g) * 64 + l) * 4 + e] =
                            in < dh ? W1[(int64_t)(32 * nb + l % 32) * dh + in] : 0.0f;
                    }
        for (int nb = 0; nb < NB1; nb++)
            for (int g = 0; g < Wd / 8; g++)
                for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 4; e++)
                        A2[((int64_t)(nb * (Wd / 8) + g) * 64 + l) * 4 + e] =
                            W2[(int64_t)(32 * nb + l % 32) * Wd + unit(4 * g + e, l / 32)];
        for (int nb = 0; nb < NB3; nb++)
            for (int g = 0; g < Wd / 8; g++)
                for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 4; e++) {
                        const int o = out_row(nb, l % 32);
                        A3[((int64_t)(nb * (Wd / 8) + g) * 64 + l) * 4 + e] =
                            o >= 0 ? W3[(int64_t)o * Wd + unit(4 * g + e, l / 32)] : 0.0f;
                    }
    }
    return ASMC_OK;
}

template <int H, int W, typename XT>
static int launch_flow(asmc_ctx* ctx, int64_t n, const XT* x, const asmc_coupling* f, double* out, hipStream_t st) {
    using FD = FlowDims<H, W>;
    const size_t all = (size_t)f->n_layers * FD::LAYER * sizeof(float);
    const int resident = all <= 150 * 1024;
    const size_t lds = resident ? all : (size_t)FD::LAYER * sizeof(float);
    ASMC_REQUIRE(lds <= 160 * 1024, "one coupling layer does not fit in LDS");
    auto kern = k_coupling_logprob<H, W, XT>;
    static size_t attr_lds = 0;  // per instantiation
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t want = (n_tiles + FLOW_WAVES - 1) / FLOW_WAVES;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;  // blocks that fit a CU's LDS
    const int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);  // -d/2 log(2 pi)
    ASMC_LAUNCH(ctx, st, "k_coupling_logprob", kern, dim3(grid), dim3(FLOW_THREADS), lds, st, n, (int)f->dims, x,
                f->packed_dev, (int)f->n_layers, resident, f->loc_dev, f->scale_dev, ladj0, base_const, out);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

template <typename XT>
static int dispatch_flow(asmc_ctx* ctx, int64_t n, const XT* x, const asmc_coupling* f, double* out, hipStream_t st) {
    const int H = flow_half_pad(f->dims);
#define ASMC_FLOW_CASE(HH, WW) \
    if (H == HH && f->hidden == WW) return launch_flow<HH, WW, XT>(ctx, n, x, f, out, st);
    ASMC_FLOW_CASE(16, 32)
    ASMC_FLOW_CASE(16, 64)
    ASMC_FLOW_CASE(16, 128)
    ASMC_FLOW_CASE(32, 32)
    ASMC_FLOW_CASE(32, 64)
    ASMC_FLOW_CASE(32, 128)
#undef ASMC_FLOW_CASE
    asmc_set_error("asmc_coupling_logprob: unsupported flow shape");
    return ASMC_ERR_UNSUPPORTED;
}

extern "C" int asmc_coupling_logprob(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x_dev,
                                     const asmc_coupling* flow, double* out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(flow && x_dev && out_dev, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    ASMC_REQUIRE(flow->packed_dev && flow->loc_dev && flow->scale_dev, "flow parameters missing");
    if (!flow_supported(flow->dims, flow->hidden) || flow->n_layers < 1) {
        asmc_set_error("asmc_coupling_logprob: unsupported flow (dims even and <= 64, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == ASMC_F64) return dispatch_flow<double>(ctx, n, (const double*)x_dev, flow, out_dev, st);
    if (x_dtype == ASMC_F32) return dispatch_flow<float>(ctx, n, (const float*)x_dev, flow, out_dev, st);
    asmc_set_error("asmc_coupling_logprob: bad x_dtype");
    return ASMC_ERR_ARG;
}
