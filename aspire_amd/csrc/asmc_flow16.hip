// asmc_flow16.hip — neural-flow proposals at 32 < d <= 128 (round 5): packing, the stand-alone density kernel and the
// ONE-kernel flow-proposal pCN / tpCN step on 16-particle groups (asmc_flow16_dev.h).
//
// Reference: the tempered target of every MCMC step evaluates the proposal flow's density at any dims
// (src/aspire/samplers/smc/base.py:507-519 around flows/torch/flows.py:368-387); the reference's default flow class is a
// masked autoregressive flow (flows/torch/flows.py:140).  Rounds 2-4 had a one-kernel step at d <= 32 only; a 64-dimensional
// coupling flow ran propose / flow / targets / accept / copy kernels (1.2 ms per step at 1M particles) and an autoregressive
// flow left the device above 32 dimensions.  Here the step of asmc_pcn_mm.hip (whitened state, x' = mu + L y' on the fp64
// matrix cores, four lanes per particle) evaluates the flow between its proposal and its accept test, with the flow's
// weights streamed through two LDS slots (asmc_flow16_dev.h).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "asmc_common.h"
#include "asmc_pcn_dev.h"
#include "asmc_flow16_dev.h"

typedef double doublex4_f16 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------------
// layout decision: which packed layout a flow of this shape has (pure function of the shape)
static int f16_pad_dim(int dims) { return dims <= 64 ? 64 : 128; }

extern "C" int asmc_flow_layout(int kind, int dims, int hidden) {
    if (!(hidden == 32 || hidden == 64 || hidden == 128)) return -1;
    if (kind == ASMC_FLOW_COUPLING) {
        if (dims < 2 || dims % 2 || dims > 128) return -1;
        return dims > 32 ? 1 : 0;
    }
    if (kind == ASMC_FLOW_MAF) {
        if (dims < 1 || dims > 128) return -1;
        // hidden width 128 at d <= 32 (round 6): the d <= 32 kernels keep every transform resident in LDS - 113 KB each at this width, so
        // two transforms already do not fit (round 5 refused them: "do not fit in LDS together") - and their one-kernel step has no
        // autoregressive instantiation of this width.  The 16-particle-group kernels stream their weights: the flow runs on them,
        // zero-padded to D = 64 (density, sampling and the one-kernel step).
        return (dims > 32 || hidden == 128) ? 1 : 0;
    }
    return -1;
}

// padded row r (of D) -> natural coordinate of the d-dimensional problem, or -1 for padding.  A coupling flow splits x at
// d / 2; the padded layout keeps each half at the start of its half of D, so that a lane's first D / 8 slots are conditioner
// inputs and the others transformed coordinates whatever d is.  An autoregressive flow keeps the natural order.
__host__ __device__ static inline int f16_nat(int kind, int D, int d, int r) {
    if (kind == ASMC_FLOW_MAF) return r < d ? r : -1;
    const int dh = d / 2;
    if (r < D / 2) return r < dh ? r : -1;
    const int j = r - D / 2;
    return j < dh ? dh + j : -1;
}

static inline uint16_t f16_bits(_Float16 v) {
    uint16_t b;
    memcpy(&b, &v, 2);
    return b;
}

template <int KIND, int D, int W>
static int pack16(int dims, int n_layers, const float* const* weights_host, const float* const* biases_host, float* packed_host) {
    using FD = Flow16<KIND, D, W>;
    constexpr bool MAF = FD::MAF;
    const int dh = MAF ? dims : dims / 2;  // inputs of the first layer = transformed coordinates
    const int64_t sizes[3] = {(int64_t)W * dh, (int64_t)W * W, (int64_t)2 * dh * W};
    for (int c = 0; c < 3 * n_layers; c++)
        for (int64_t k = 0; k < sizes[c % 3]; k++)
            if (!(fabsf(weights_host[c][k]) < 65504.0f)) {
                asmc_set_error("flow pack: weight %g of matrix %d is outside the fp16 operand range of the split-fp16 flow kernels (|w| < 65504)",
                               (double)weights_host[c][k], c);
                return ASMC_ERR_UNSUPPORTED;
            }
    // index of the conditioner input / transformed coordinate held in slot s of lane group g -> column / row of the torch matrices
    // (coupling: index inside the half; autoregressive: the coordinate), or -1 for padding
    auto cidx = [&](int s, int g) {
        const int c = f16_coord(s, g);  // coupling: s < CS, index inside the padded half; MAF: padded coordinate
        return c < dh ? c : -1;
    };
    float* bias = packed_host;
    uint16_t* img = reinterpret_cast<uint16_t*>(packed_host + (int64_t)n_layers * FD::BIAS);
    auto put = [&](uint16_t* blockimg, int KS, int S, int lane, int j, float w) {
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        // [K step][hi | lo][lane][8 halves]
        blockimg[((size_t)(S * 2 + 0) * 64 + lane) * 8 + j] = f16_bits(hi);
        blockimg[((size_t)(S * 2 + 1) * 64 + lane) * 8 + j] = f16_bits(lo);
        (void)KS;
    };
    for (int c = 0; c < n_layers; c++) {
        const float *W1 = weights_host[3 * c], *W2 = weights_host[3 * c + 1], *W3 = weights_host[3 * c + 2];
        const float *B1 = biases_host[3 * c], *B2 = biases_host[3 * c + 1], *B3 = biases_host[3 * c + 2];
        float* b = bias + (int64_t)c * FD::BIAS;
        for (int u = 0; u < W; u++) b[u] = B1[u], b[W + u] = B2[u];
        // output row (block mbo, lane group g, register r): block 2 m = s_raw, 2 m + 1 = t of the lane's transformed slots 4 m + r
        auto out_row = [&](int mbo, int g, int r) {
            const int i = cidx(4 * (mbo / 2) + r, g);
            if (i < 0) return -1;
            return (mbo & 1) ? dh + i : i;
        };
        for (int mbo = 0; mbo < FD::NB3; mbo++)
            for (int g = 0; g < 4; g++)
                for (int r = 0; r < 4; r++) {
                    const int o = out_row(mbo, g, r);
                    b[2 * W + (mbo * 4 + g) * 4 + r] = o >= 0 ? B3[o] : 0.0f;
                }
        uint16_t* A1 = img + (size_t)c * FD::LAYER_A * 2;
        uint16_t* A2 = A1 + (size_t)FD::A1 * 2;
        uint16_t* A3 = A2 + (size_t)FD::A2 * 2;
        for (int mbo = 0; mbo < FD::NB1; mbo++)
            for (int S = 0; S < FD::KS1; S++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int m = lane & 15, kg = lane >> 4;
                        const int in = cidx(8 * S + j, kg);
                        put(A1 + (size_t)mbo * FD::BLK1 * 2, FD::KS1, S, lane, j, in >= 0 ? W1[(int64_t)(16 * mbo + m) * dh + in] : 0.0f);
                    }
        for (int mbo = 0; mbo < FD::NB1; mbo++)
            for (int S = 0; S < FD::KS2; S++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int m = lane & 15, kg = lane >> 4;
                        const int u = 16 * (2 * S + j / 4) + 4 * kg + j % 4;
                        put(A2 + (size_t)mbo * FD::BLK2 * 2, FD::KS2, S, lane, j, W2[(int64_t)(16 * mbo + m) * W + u]);
                    }
        for (int mbo = 0; mbo < FD::NB3; mbo++)
            for (int S = 0; S < FD::KS2; S++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int m = lane & 15, kg = lane >> 4;
                        const int u = 16 * (2 * S + j / 4) + 4 * kg + j % 4;
                        const int o = out_row(mbo, m >> 2, m & 3);
                        put(A3 + (size_t)mbo * FD::BLK2 * 2, FD::KS2, S, lane, j, o >= 0 ? W3[(int64_t)o * W + u] : 0.0f);
                    }
    }
    return ASMC_OK;
}

#define F16_SHAPES(X)                                                          \
    X(ASMC_FLOW_COUPLING, 64, 32) X(ASMC_FLOW_COUPLING, 64, 64) X(ASMC_FLOW_COUPLING, 64, 128)    \
    X(ASMC_FLOW_COUPLING, 128, 32) X(ASMC_FLOW_COUPLING, 128, 64) X(ASMC_FLOW_COUPLING, 128, 128) \
    X(ASMC_FLOW_MAF, 64, 32) X(ASMC_FLOW_MAF, 64, 64) X(ASMC_FLOW_MAF, 64, 128)                   \
    X(ASMC_FLOW_MAF, 128, 32) X(ASMC_FLOW_MAF, 128, 64) X(ASMC_FLOW_MAF, 128, 128)

int64_t asmc_flow16_pack_floats(int kind, int dims, int n_layers, int hidden) {
    const int D = f16_pad_dim(dims);
#define X(K, DD, WW) \
    if (kind == K && D == DD && hidden == WW) return flow16_words<K, DD, WW>(n_layers);
    F16_SHAPES(X)
#undef X
    return ASMC_ERR_UNSUPPORTED;
}

int asmc_flow16_pack(int kind, int dims, int n_layers, int hidden, const float* const* weights_host, const float* const* biases_host,
                     float* packed_host) {
    const int D = f16_pad_dim(dims);
#define X(K, DD, WW) \
    if (kind == K && D == DD && hidden == WW) return pack16<K, DD, WW>(dims, n_layers, weights_host, biases_host, packed_host);
    F16_SHAPES(X)
#undef X
    asmc_set_error("flow pack: unsupported shape (kind %d, dims %d, hidden %d)", kind, dims, hidden);
    return ASMC_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------------
// stand-alone density: out[i] = log q(x_i), x row-major [n, d]
template <int KIND, int D, int W, typename XT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_flow16_logprob(int64_t n, int d, const XT* __restrict__ x, const float* __restrict__ packed,
                                                           int n_layers, const float* __restrict__ loc, const float* __restrict__ scale,
                                                           float ladj0, float base_const, double* __restrict__ out, int form) {
    using FD = Flow16<KIND, D, W>;
    extern __shared__ __align__(16) float sm[];
    constexpr int WAVES = THREADS / 64, SL = FD::SL;
    float* slots = sm;                                  // 2 x FLOW16_CHUNK_WORDS
    float* s_bias = slots + 2 * FD::CW;                 // n_layers x BIAS
    float* s_loc = s_bias + n_layers * FD::BIAS;        // [slot s][lane group h] x {loc, scale, 1/scale}: 3 x D floats
    Flow16Stream<FD, THREADS> stream;
    stream.start(packed + (size_t)n_layers * FD::BIAS, slots, n_layers);
    for (int e = threadIdx.x; e < n_layers * FD::BIAS; e += THREADS) s_bias[e] = packed[e];
    for (int e = threadIdx.x; e < D; e += THREADS) {
        const int s = e >> 2, h = e & 3, j = f16_nat(KIND, D, d, f16_coord(s, h));
        s_loc[e] = j >= 0 ? loc[j] : 0.0f;
        s_loc[D + e] = j >= 0 ? scale[j] : 1.0f;
        s_loc[2 * D + e] = j >= 0 ? 1.0f / scale[j] : 1.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, h = lane >> 4;
    const int64_t n_groups = (n + 15) / 16;
    const int64_t per_round = (int64_t)gridDim.x * WAVES;
    const int64_t rounds = (n_groups + per_round - 1) / per_round;  // the same trip count for every wave: they share the stream's barriers
    for (int64_t it = 0; it < rounds; it++) {
        const int64_t g = it * per_round + (int64_t)blockIdx.x * WAVES + wave;
        const int64_t row = g * 16 + p;
        const bool valid = row < n;
        F16State<FD> xf;
#pragma unroll
        for (int s = 0; s < SL; s++) {
            const int j = f16_nat(KIND, D, d, f16_coord(s, h));
            const float xv = (valid && j >= 0) ? (float)x[row * d + j] : 0.0f;
            xf.set(s, j >= 0 ? flow_standardise(xv, s_loc[s * 4 + h], s_loc[D + s * 4 + h], s_loc[2 * D + s * 4 + h]) : 0.0f);
        }
        // (the affine form as a compile-time constant of each branch: asmc_flow16_dev.h f16_layer)
        const float val = (!FD::MAF || form == 0) ? f16_logprob<FD, W, THREADS, 0>(xf, n_layers, s_bias, stream, lane, ladj0, base_const, 0)
                                                  : f16_logprob<FD, W, THREADS, 1>(xf, n_layers, s_bias, stream, lane, ladj0, base_const, 1);
        if (valid && h == 0) out[row] = (double)val;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk the last next() put into flight
}

template <int KIND, int D, int W, typename XT>
static int launch_flow16_logprob(asmc_ctx* ctx, int64_t n, const XT* x, const asmc_coupling* f, double* out, hipStream_t st) {
    using FD = Flow16<KIND, D, W>;
    constexpr int THREADS = 512;
    const size_t lds = (size_t)(2 * FLOW16_CHUNK_WORDS + f->n_layers * FD::BIAS + 3 * D) * sizeof(float);
    ASMC_REQUIRE(lds <= 160 * 1024, "flow16: biases exceed the LDS");
    auto kern = k_flow16_logprob<KIND, D, W, XT, THREADS>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t n_groups = (n + 15) / 16;
    const int64_t want = (n_groups + THREADS / 64 - 1) / (THREADS / 64);
    const int per_cu = 2;  // 68 KB of LDS per block
    const int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    ASMC_LAUNCH(ctx, st, "k_flow16_logprob", kern, dim3(grid), dim3(THREADS), lds, st, n, (int)f->dims, x, f->packed_dev, (int)f->n_layers,
                f->loc_dev, f->scale_dev, ladj0, base_const, out, (int)f->affine);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_flow16_logprob(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x, const asmc_coupling* f, double* out, hipStream_t st) {
    if (!asmc_flow_math_split()) {
        asmc_set_error("flows of more than 32 dimensions run the split-fp16 layers only (ASMC_FLOW_MATH=f32 is not available there)");
        return ASMC_ERR_UNSUPPORTED;
    }
    const int D = f16_pad_dim(f->dims);
#define X(K, DD, WW)                                                                                                              \
    if (f->kind == K && D == DD && f->hidden == WW) {                                                                             \
        if (x_dtype == ASMC_F64) return launch_flow16_logprob<K, DD, WW, double>(ctx, n, (const double*)x, f, out, st);          \
        return launch_flow16_logprob<K, DD, WW, float>(ctx, n, (const float*)x, f, out, st);                                     \
    }
    F16_SHAPES(X)
#undef X
    asmc_set_error("flow16: unsupported shape (kind %d, dims %d, hidden %d)", (int)f->kind, (int)f->dims, (int)f->hidden);
    return ASMC_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The one-kernel flow-proposal step (pCN / tpCN) for D = 64 / 128: asmc_pcn_mm.hip's MM_STEP with the flow's density in the
// place of the built-in log q.
//
// Resident tables, built once per mutation by k_flow16_tables in exactly the order the kernel keeps them in LDS (doubles):
//   sA    [mm_ksum(D / 16) * 64]  MFMA A-operand image of L with its ROWS in the flow's padded order (f16_nat): x' comes out
//                                 with each half of a coupling flow at the start of its half of D; y keeps the natural order
//   s_mu  [D]                     reference mean in lane order, rows as above
//   t_ll, t_lp [C x 2 D] each     the built-in targets' (mean, precision) tables in lane order, rows as above
//   s_loc [3 D floats]            the flow's loc / scale / 1 / scale in lane order [slot][lane group]
__host__ __device__ constexpr int f16_ksum(int nb) { return 2 * nb * (nb + 1); }

__global__ __launch_bounds__(256) void k_flow16_tables(int kind, int D, int d, const double* __restrict__ L, const double* __restrict__ mu,
                                                      MixDev ll, MixDev lp, const float* __restrict__ loc, const float* __restrict__ scale,
                                                      double* __restrict__ blob) {
    const int nb = D / 16, total = f16_ksum(nb) * 64;
    double* sA = blob;
    double* s_mu = sA + total;
    double* t_ll = s_mu + D;
    double* t_lp = t_ll + (size_t)ll.C * D * 2;
    float* s_loc = reinterpret_cast<float*>(t_lp + (size_t)lp.C * D * 2);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, ks = e >> 6;
        int ib = 0;
        while (f16_ksum(ib + 1) <= ks) ib++;
        const int s = ks - f16_ksum(ib);
        const int rho = lane & 15, kk = lane >> 4;
        const int i = f16_nat(kind, D, d, f16_coord(4 * ib + rho / 4, rho % 4)), k = f16_coord(s, kk);
        sA[e] = (i >= 0 && k <= i) ? L[(size_t)i * D + k] : 0.0;
    }
    if (blockIdx.x != 0) return;
    for (int e = threadIdx.x; e < D; e += 256) {
        const int q = e & 1, hh = (e >> 1) & 3, sp = e >> 3;
        const int i = f16_nat(kind, D, d, f16_coord(2 * sp + q, hh));
        s_mu[e] = i >= 0 ? mu[i] : 0.0;
    }
    for (int t = 0; t < 2; t++) {
        const MixDev& m = t == 0 ? ll : lp;
        double* tab = t == 0 ? t_ll : t_lp;
        const int per_c = D * 2;
        for (int e = threadIdx.x; e < m.C * per_c; e += 256) {
            const int c = e / per_c, w = e - c * per_c;
            const int q = w & 3, hh = (w >> 2) & 3, sp = w >> 4;
            const int i = f16_nat(kind, D, d, f16_coord(2 * sp + (q & 1), hh));
            tab[e] = i < 0 ? 0.0 : (q < 2) ? m.mu[(size_t)c * D + i] : m.prec[(size_t)c * D + i];
        }
    }
    for (int e = threadIdx.x; e < D; e += 256) {
        const int s = e >> 2, h = e & 3, j = f16_nat(kind, D, d, f16_coord(s, h));
        s_loc[e] = j >= 0 ? loc[j] : 0.0f;
        s_loc[D + e] = j >= 0 ? scale[j] : 1.0f;
        s_loc[2 * D + e] = j >= 0 ? 1.0f / scale[j] : 1.0f;
    }
}

template <int D>
__device__ __forceinline__ void f16_trimatvec(const double* __restrict__ sA, const double (&v)[D / 4], double (&out)[D / 4], int lane) {
    constexpr int NB = D / 16;
#pragma unroll
    for (int pr = 0; pr < NB / 2; pr++) {
        const int ia = pr, ib = NB - 1 - pr;
        doublex4_f16 acc_a = {0.0, 0.0, 0.0, 0.0}, acc_b = {0.0, 0.0, 0.0, 0.0};
        const double* Aa = sA + (size_t)f16_ksum(ia) * 64 + lane;
        const double* Ab = sA + (size_t)f16_ksum(ib) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 4 * ib + 4; s++) {
            if (s < 4 * ia + 4) acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(Aa[(size_t)s * 64], v[s], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[(size_t)s * 64], v[s], acc_b, 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (operand reads stay with their K steps: asmc_pcn_mm.hip)
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            out[4 * ia + r] = acc_a[r];
            out[4 * ib + r] = acc_b[r];
        }
    }
}

__device__ __forceinline__ double f16_quad_sum_d(double q) {
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    return q;
}

// diagonal-mixture log-density from the lane's coordinates (asmc_pcn_mm.hip mm_mixture: same order of operations)
template <int D>
__device__ __forceinline__ double f16_mixture(const MixDev& m, const double* __restrict__ tab, const double* __restrict__ logw,
                                              const double (&xv)[D / 4], int h) {  // logw: the log-weights, staged in LDS (see mm_mixture)
    double terms[ASMC_MAX_COMPONENTS];
    double best = -INFINITY;
    for (int c = 0; c < m.C; c++) {
        const double* tc = tab + (size_t)c * D * 2 + h * 4;
        double q = 0.0;
#pragma unroll
        for (int sp = 0; sp < D / 8; sp++) {
            const double2 mu2 = *reinterpret_cast<const double2*>(tc + sp * 16);
            const double2 pr2 = *reinterpret_cast<const double2*>(tc + sp * 16 + 2);
            const double t0 = xv[2 * sp] - mu2.x, t1 = xv[2 * sp + 1] - mu2.y;
            q = fma(t0 * t0, pr2.x, q);
            q = fma(t1 * t1, pr2.y, q);
        }
        q = f16_quad_sum_d(q);
        terms[c] = logw[c] - 0.5 * q;
        best = fmax(best, terms[c]);
    }
    if (m.C == 1) return terms[0];
    if (!(best > -INFINITY)) return best;
    double sum = 0.0;
    for (int c = 0; c < m.C; c++) sum += exp(terms[c] - best);
    return best + log(sum);
}

#ifdef F16_SKEW
__device__ unsigned g_f16_ticket[4096];
#endif
template <typename T, int D, int W, int KIND, int NOISE, bool TP, int THREADS, int CW>
__global__ __launch_bounds__(THREADS, THREADS > 512 ? 1 : 2) void k_pcn_flow16(int64_t n, T* __restrict__ x, double* __restrict__ ll, double* __restrict__ lp,
                                                       double* __restrict__ lq, const double* __restrict__ blob, int blob_doubles,
                                                       PcnDev p, const double* __restrict__ rho_ptr, uint32_t step,
                                                       const float* __restrict__ packed, int n_layers, float ladj0, float base_const,
                                                       long long* __restrict__ block_counts, unsigned long long* __restrict__ nonfinite, int form) {
    using FD = Flow16<KIND, D, W, CW>;
    extern __shared__ __align__(16) double smem[];
    constexpr int KS = D / 4, WAVES = THREADS / 64;
    constexpr int TOTAL = f16_ksum(D / 16) * 64;
    double* sA = smem;
    double* s_mu = sA + TOTAL;
    double* t_ll = s_mu + D;
    double* t_lp = t_ll + (size_t)p.ll.C * D * 2;
    float* s_loc = reinterpret_cast<float*>(t_lp + (size_t)p.lp.C * D * 2);
    float* s_bias = s_loc + 3 * D;
    bm_d2* bmt = reinterpret_cast<bm_d2*>(s_bias + n_layers * FD::BIAS);
    float* slots = reinterpret_cast<float*>(bmt + (NOISE == ASMC_NOISE_F64 ? BM_TAB_N : 0));
    Flow16Stream<FD, THREADS> stream;
    stream.start(packed + (size_t)n_layers * FD::BIAS, slots, n_layers);
    for (int e = threadIdx.x * 2; e < blob_doubles; e += THREADS * 2) *reinterpret_cast<double2*>(smem + e) = *reinterpret_cast<const double2*>(blob + e);
    for (int e = threadIdx.x; e < n_layers * FD::BIAS; e += THREADS) s_bias[e] = packed[e];
    if (NOISE == ASMC_NOISE_F64) bm_tab_stage<THREADS>(bmt, p.bmtab);
    __shared__ double s_logw[2 * ASMC_MAX_COMPONENTS];  // log-weights of (ll, lp): read per group, never through the global pointer
    if (threadIdx.x < 2 * ASMC_MAX_COMPONENTS) {
        const int c = threadIdx.x % ASMC_MAX_COMPONENTS;
        const MixDev& mt = threadIdx.x < ASMC_MAX_COMPONENTS ? p.ll : p.lp;
        s_logw[threadIdx.x] = c < mt.C ? mt.logw[c] : 0.0;
    }
    __syncthreads();
#ifdef F16_SKEW  // diagnostic builds (two 4-wave blocks per CU): the SECOND block to arrive on a CU starts F16_SKEW x 8128 cycles late
    {
        __shared__ unsigned s_tick;
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            s_tick = atomicAdd(&g_f16_ticket[((xcc & 15u) << 8) | ((hw >> 8) & 0xFFu)], 1u);
        }
        __syncthreads();
        if (s_tick & 1u)
            for (int i = 0; i < F16_SKEW; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pp = lane & 15, h = lane >> 4;
    const int dn = (p.d_noise > 0 && p.d_noise < D) ? p.d_noise : D;  // real dimension of a zero-padded problem
    const double rho = *rho_ptr;
    const double a = sqrt(1.0 - rho * rho);
    const int64_t n_groups = (n + 15) / 16;
    const int64_t per_round = (int64_t)gridDim.x * WAVES;
    const int64_t rounds = (n_groups + per_round - 1) / per_round;  // every wave of a block meets the same barriers
    long long n_acc = 0, n_bad = 0;
    struct alignas(2 * sizeof(T)) Pair {
        T a, b;
    };
    for (int64_t it = 0; it < rounds; it++) {
        const int64_t g = it * per_round + (int64_t)blockIdx.x * WAVES + wave;
        const int64_t row = g * 16 + pp;
        const bool valid = row < n;
        Pair* xr = reinterpret_cast<Pair*>(x + (valid ? row : 0) * D + 2 * h);  // owned pairs sit 8 elements apart
        double v[KS], o[KS];
#pragma unroll
        for (int sp = 0; sp < KS / 2; sp++) {
            Pair t = {(T)0, (T)0};
            if (valid) t = xr[sp * 4];
            v[2 * sp] = (double)t.a;
            v[2 * sp + 1] = (double)t.b;
        }
        double oll = 0.0, olp = 0.0, olq = 0.0;
        if (valid) oll = ll[row], olp = lp[row], olq = lq[row];
        const unsigned long long gid = p.gid0 + (unsigned long long)(valid ? row : 0);
        double q0 = 0.0, q1 = 0.0;
#pragma unroll
        for (int s = 0; s < KS; s++) q0 = fma(v[s], v[s], q0);
        q0 = f16_quad_sum_d(q0);
        const double rs = tpcn_scale_ct<TP>(rho, p.nu, q0, p.gam, valid ? row : 0);
        // noise: asmc_pcn_mm.hip's sharing of Philox blocks between the lanes of a pair (same normals)
        auto swap64 = [](double& xx, double& yy) {  // odd rows of xx <-> even rows of yy
            const unsigned long long xb = (unsigned long long)__double_as_longlong(xx), yb = (unsigned long long)__double_as_longlong(yy);
            const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
            xx = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
            yy = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
        };
#pragma unroll
        for (int m = 0; m < KS / 4; m++) {
            double zz[4];
            uint32_t blk = (uint32_t)(2 * (2 * m + (h & 1)) + (h >> 1));
            // D = 128 sits at the register limit: left alone, the compiler hoists the parts of each block's first Philox round that do
            // not change from round to round out of the loop over the rounds - eight blocks' worth of lane-dependent values, which it
            // then spills and reloads with a full wait in front of every block.  An opaque block index keeps them inside the loop.
            if constexpr (D == 128) asm volatile("" : "+v"(blk));
            if (NOISE == ASMC_NOISE_F32) {
                float f0, f1, f2, f3;
                normal_quad_f32_raw(p.seed, gid, step, blk, f0, f1, f2, f3);
                const auto s02 = __builtin_amdgcn_permlane16_swap(__float_as_uint(f0), __float_as_uint(f2), false, false);
                const auto s13 = __builtin_amdgcn_permlane16_swap(__float_as_uint(f1), __float_as_uint(f3), false, false);
                zz[0] = (double)__uint_as_float(s02[0]), zz[1] = (double)__uint_as_float(s13[0]);
                zz[2] = (double)__uint_as_float(s02[1]), zz[3] = (double)__uint_as_float(s13[1]);
            } else {
                normal_quad(p.seed, gid, step, blk, bmt, zz[0], zz[1], zz[2], zz[3]);
                swap64(zz[0], zz[2]);
                swap64(zz[1], zz[3]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int s = 4 * m + e;
                v[s] = (double)(T)fma(rs, zz[e], a * v[s]);
                q1 = fma(v[s], v[s], q1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dn < D) {  // a zero-padded problem: the padded coordinates carry no noise, y' = 0 there
            q1 = 0.0;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                v[s] = f16_coord(s, h) < dn ? v[s] : 0.0;
                q1 = fma(v[s], v[s], q1);
            }
        }
        q1 = f16_quad_sum_d(q1);
        f16_trimatvec<D>(sA, v, o, lane);
        const double* my_mu = s_mu + h * 2;
#pragma unroll
        for (int sp = 0; sp < KS / 2; sp++) {
            const double2 m2 = *reinterpret_cast<const double2*>(my_mu + sp * 8);
            o[2 * sp] = (double)(T)(m2.x + o[2 * sp]);
            o[2 * sp + 1] = (double)(T)(m2.y + o[2 * sp + 1]);
        }
        const double nll = f16_mixture<D>(p.ll, t_ll, s_logw, o, h), nlp = f16_mixture<D>(p.lp, t_lp, s_logw + ASMC_MAX_COMPONENTS, o, h);
        F16State<FD> xf;
#pragma unroll
        for (int s = 0; s < KS; s++) xf.set(s, flow_standardise((float)o[s], s_loc[s * 4 + h], s_loc[D + s * 4 + h], s_loc[2 * D + s * 4 + h]));
        // everything of the accept test that does not need log q(x'), pinned in front of the flow
        const double c1 = ref_corr_ct<TP>(q1, p.nu, dn), c0 = ref_corr_ct<TP>(q0, p.nu, dn);
        const double logu = bm_log_unit(accept_uniform(p.seed, gid, step));  // (< 1 ulp for the uniform's range: asmc_pcn_fused.hip)
        const double lpo = log_p_t(oll, olp, olq, p.beta);
        // ... folded into ONE number, as the d <= 32 step does (asmc_pcn_fused.hip): accept  <=>  log u < ((1 - beta) lq' + t2 + c1) - k_old
        // <=>  (1 - beta) lq' + kacc > 0 - three doubles alive across the flow instead of five (D = 128 spills them); the
        // re-association moves a decision only where its margin is within an ulp or two (the razor edges the parity tests allow for).
        // A +inf or NaN target term turns kacc into NaN: rejected (log_p_t's rule).
        double t2 = p.beta * (nll + nlp);
        t2 = (t2 < INFINITY) ? t2 : __builtin_nan("");
        double kacc = ((t2 + c1) - (lpo + c0)) - logu, k_ll = nll, k_lp = nlp;
        asm volatile("" : "+v"(kacc), "+v"(k_ll), "+v"(k_lp));
        __builtin_amdgcn_sched_barrier(0);
        const double nlq = (!FD::MAF || form == 0) ? (double)f16_logprob<FD, W, THREADS, 0, true>(xf, n_layers, s_bias, stream, lane, ladj0, base_const, 0)
                                                   : (double)f16_logprob<FD, W, THREADS, 1, true>(xf, n_layers, s_bias, stream, lane, ladj0, base_const, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (valid && h == 0 && !(fabs(nlq) < INFINITY)) n_bad++;
        const bool lq_finite = fabs(nlq) < INFINITY;
        if (valid && lq_finite && fma(1.0 - p.beta, nlq, kacc) > 0.0) {  // (a NaN / infinite density: rejected)
#pragma unroll
            for (int sp = 0; sp < KS / 2; sp++) {
                Pair t;
                t.a = (T)v[2 * sp];
                t.b = (T)v[2 * sp + 1];
                xr[sp * 4] = t;
            }
            if (h == 0) {
                ll[row] = k_ll, lp[row] = k_lp, lq[row] = nlq;
                n_acc++;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk the last next() put into flight
    __shared__ long long s_cnt[WAVES];
    n_acc = wave_sum_ll(n_acc);
    n_bad = wave_sum_ll(n_bad);
    if (lane == 0 && n_bad != 0 && nonfinite) atomicAdd(nonfinite, (unsigned long long)n_bad);
    if (lane == 0) s_cnt[wave] = n_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < WAVES; w++) t += s_cnt[w];
        block_counts[blockIdx.x] = t;
    }
}

static size_t f16_blob_doubles(int D, int c_ll, int c_lp) {
    return (size_t)f16_ksum(D / 16) * 64 + D + (size_t)(c_ll + c_lp) * D * 2 + (3 * D) / 2;
}

template <int KIND, int D, int W, int CW = FLOW16_CHUNK_WORDS>
static size_t f16_step_lds(int n_layers, int c_ll, int c_lp, int noise) {
    using FD = Flow16<KIND, D, W, CW>;
    return f16_blob_doubles(D, c_ll, c_lp) * 8 + (size_t)n_layers * FD::BIAS * 4 + (noise == ASMC_NOISE_F64 ? BM_TAB_N * sizeof(bm_d2) : 0) +
           2 * (size_t)CW * 4 + 16 * 8 /* s_cnt */ + 2 * ASMC_MAX_COMPONENTS * 8 /* s_logw */;
}

// shapes of the one-kernel step (every instantiation is x 2 state dtypes x 2 noise generators x pCN / tpCN)
#ifdef F16_DIAG_ONLY_D128  // diagnostic builds (register / spill experiments): the D = 128 steps alone, seconds instead of minutes
#define F16_STEP_SHAPES(X) X(ASMC_FLOW_COUPLING, 128, 64) X(ASMC_FLOW_MAF, 128, 64)
#elif defined(F16_DIAG_ONLY_D64)
#define F16_STEP_SHAPES(X) X(ASMC_FLOW_COUPLING, 64, 64)
#else
// (hidden widths 32 and 128 since round 6 - the reference forwards any `hidden_features`, flows/torch/flows.py:164; width 128 keeps
// 64 more activation registers per lane than width 64 and spills at D = 128: one kernel, but not the speed of the default width)
#define F16_STEP_SHAPES(X)                                                                                                          \
    X(ASMC_FLOW_COUPLING, 64, 64) X(ASMC_FLOW_COUPLING, 128, 64) X(ASMC_FLOW_MAF, 64, 64) X(ASMC_FLOW_MAF, 128, 64)                  \
    X(ASMC_FLOW_COUPLING, 64, 32) X(ASMC_FLOW_COUPLING, 128, 32) X(ASMC_FLOW_MAF, 64, 32) X(ASMC_FLOW_MAF, 128, 32)                  \
    X(ASMC_FLOW_COUPLING, 64, 128) X(ASMC_FLOW_COUPLING, 128, 128) X(ASMC_FLOW_MAF, 64, 128) X(ASMC_FLOW_MAF, 128, 128)
#endif

// whether the one-kernel step takes this mutation (prm->d is the PADDED dimension 64 / 128; the flow keeps its own dims)
bool asmc_pcn_flow16_ok(const asmc_pcn_params* prm, const asmc_coupling* f) {
    if (getenv("ASMC_FLOW_SPLIT") || getenv("ASMC_FLOW16_OFF") || !asmc_flow_math_split()) return false;
    if (!(prm->d == 64 || prm->d == 128) || asmc_flow_layout(f->kind, f->dims, f->hidden) != 1 || f16_pad_dim(f->dims) != prm->d) return false;
    const int cl = prm->log_likelihood.n_components, cp = prm->log_prior.n_components;
    if (cl < 1 || cp < 1 || cl > ASMC_MAX_COMPONENTS || cp > ASMC_MAX_COMPONENTS) return false;
    size_t lds = 0;
#define X(K, DD, WW) \
    if (f->kind == K && prm->d == DD && f->hidden == WW) lds = f16_step_lds<K, DD, WW>(f->n_layers, cl, cp, prm->noise);
    F16_STEP_SHAPES(X)
#undef X
    return lds > 0 && lds <= 160 * 1024;
}

// Launch geometry: one block of 8 waves per CU (two per SIMD; the step sits at ~200 registers).  CW = the LDS slot of the
// weight stream: 32 KB (one barrier per dense matrix), or - D = 64, where the L image is small - a whole layer (40 - 64 KB:
// one barrier per layer; 638 -> 600 us per step at 1M x 64, profiles/r05_flow16_geometry.txt).  Two independent 4-wave blocks
// per CU with 16 KB slots (their phases free to drift apart, vector work of one beside matrix work of the other) measured
// SLOWER, 685 us: every wave of a block is in the same phase, and two half-size blocks pay the stream twice.
template <typename T, int D, int W, int KIND, int NOISE, bool TP, int THREADS, int CW, int PER_CU>
static int launch_pcn_flow16_g(asmc_ctx* ctx, int64_t n, T* x, double* ll, double* lp, double* lq, const double* blob, const PcnDev& pd,
                               const asmc_coupling* f, const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                               unsigned long long* nonfinite, hipStream_t st) {
    const size_t lds = f16_step_lds<KIND, D, W, CW>(f->n_layers, pd.ll.C, pd.lp.C, NOISE) - 256;  // (the static part: s_cnt, s_logw)
    if ((lds + 256) * PER_CU > 160 * 1024) return ASMC_ERR_UNSUPPORTED;  // (the caller tries the next geometry)
    auto kern = k_pcn_flow16<T, D, W, KIND, NOISE, TP, THREADS, CW>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t n_groups = (n + 15) / 16;
    const int64_t want = (n_groups + THREADS / 64 - 1) / (THREADS / 64);
    const int grid = (int)(want < (int64_t)ctx->num_cu * PER_CU ? want : (int64_t)ctx->num_cu * PER_CU);
    *grid_out = grid;
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    ASMC_LAUNCH(ctx, st, TP ? "k_tpcn_flow16" : "k_pcn_flow16", kern, dim3(grid), dim3(THREADS), lds, st, n, x, ll, lp, lq, blob,
                (int)f16_blob_doubles(D, pd.ll.C, pd.lp.C), pd, rho_ptr, step, f->packed_dev, (int)f->n_layers, ladj0, base_const, block_counts,
                nonfinite, (int)f->affine);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

template <typename T, int D, int W, int KIND, int NOISE, bool TP>
static int launch_pcn_flow16(asmc_ctx* ctx, int64_t n, T* x, double* ll, double* lp, double* lq, const double* blob, const PcnDev& pd,
                             const asmc_coupling* f, const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                             unsigned long long* nonfinite, hipStream_t st) {
    static const bool small_slots = getenv("ASMC_F16_SMALL_SLOTS") != nullptr;  // (A/B switch: 32 KB slots at D = 64 too)
    int rc = ASMC_ERR_UNSUPPORTED;
#define F16_ARGS ctx, n, x, ll, lp, lq, blob, pd, f, rho_ptr, step, block_counts, grid_out, nonfinite, st
#ifdef F16_SKEW
    if constexpr (D == 64) return launch_pcn_flow16_g<T, D, W, KIND, NOISE, TP, 256, 4096, 2>(F16_ARGS);
#endif
#ifdef F16_WAVES12  // diagnostic builds: three waves per SIMD (168 registers)
    if constexpr (D == 64) return launch_pcn_flow16_g<T, D, W, KIND, NOISE, TP, 768, Flow16<KIND, D, W>::LAYER_A, 1>(F16_ARGS);
#endif
    if constexpr (D == 64) {
        if (!small_slots) rc = launch_pcn_flow16_g<T, D, W, KIND, NOISE, TP, 512, Flow16<KIND, D, W>::LAYER_A, 1>(F16_ARGS);
    }
    if (rc == ASMC_ERR_UNSUPPORTED) rc = launch_pcn_flow16_g<T, D, W, KIND, NOISE, TP, 512, FLOW16_CHUNK_WORDS, 1>(F16_ARGS);
#undef F16_ARGS
    if (rc == ASMC_ERR_UNSUPPORTED) asmc_set_error("flow16 step: the step's tables exceed the LDS");
    return rc;
}

// builds the resident tables of a mutation (once, before its steps) into ctx->d_f16tab
int asmc_pcn_flow16_tables(asmc_ctx* ctx, const PcnDev& pd, const asmc_coupling* f, hipStream_t st) {
    const int D = pd.d;
    const size_t need = f16_blob_doubles(D, pd.ll.C, pd.lp.C) * 8 + 64;
    if (need > ctx->f16tab_bytes) {
        ASMC_HIP(hipStreamSynchronize(st));
        if (ctx->d_f16tab) (void)hipFree(ctx->d_f16tab);
        ctx->d_f16tab = nullptr, ctx->f16tab_bytes = 0;
        const size_t cap = f16_blob_doubles(128, ASMC_MAX_COMPONENTS, ASMC_MAX_COMPONENTS) * 8 + 64;
        if (hipMalloc((void**)&ctx->d_f16tab, cap) != hipSuccess) {
            (void)hipGetLastError();
            asmc_set_error("flow16: no device memory for the step's tables");
            return ASMC_ERR_NOMEM;
        }
        ctx->f16tab_bytes = cap;
    }
    const int total = f16_ksum(D / 16) * 64;
    ASMC_LAUNCH(ctx, st, "k_flow16_tables", k_flow16_tables, dim3((total + 255) / 256), dim3(256), 0, st, (int)f->kind, D,
                pd.d_noise > 0 ? pd.d_noise : D, pd.L, pd.mu, pd.ll, pd.lp, f->loc_dev, f->scale_dev, ctx->d_f16tab);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_pcn_flow16_launch(asmc_ctx* ctx, int64_t n, int x_dtype, void* x, double* ll, double* lp, double* lq, const PcnDev& pd,
                           const asmc_coupling* f, const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                           unsigned long long* nonfinite, hipStream_t st) {
    const bool tp = pd.nu > 0.0;
#define F16_CASE(TT, K, DD, WW, NZ, TPV) \
    return launch_pcn_flow16<TT, DD, WW, K, NZ, TPV>(ctx, n, (TT*)x, ll, lp, lq, ctx->d_f16tab, pd, f, rho_ptr, step, block_counts, grid_out, nonfinite, st);
#define X(K, DD, WW)                                                                                        \
    if (f->kind == K && pd.d == DD && f->hidden == WW) {                                                    \
        if (x_dtype == ASMC_F64) {                                                                          \
            if (pd.noise == ASMC_NOISE_F32) { if (tp) { F16_CASE(double, K, DD, WW, ASMC_NOISE_F32, true) } F16_CASE(double, K, DD, WW, ASMC_NOISE_F32, false) } \
            if (tp) { F16_CASE(double, K, DD, WW, ASMC_NOISE_F64, true) }                                   \
            F16_CASE(double, K, DD, WW, ASMC_NOISE_F64, false)                                              \
        }                                                                                                   \
        if (pd.noise == ASMC_NOISE_F32) { if (tp) { F16_CASE(float, K, DD, WW, ASMC_NOISE_F32, true) } F16_CASE(float, K, DD, WW, ASMC_NOISE_F32, false) } \
        if (tp) { F16_CASE(float, K, DD, WW, ASMC_NOISE_F64, true) }                                        \
        F16_CASE(float, K, DD, WW, ASMC_NOISE_F64, false)                                                   \
    }
    F16_STEP_SHAPES(X)
#undef X
#undef F16_CASE
    asmc_set_error("flow16 step: unsupported shape (kind %d, dims %d, hidden %d)", (int)f->kind, (int)f->dims, (int)f->hidden);
    return ASMC_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Sampling (flows/torch/flows.py:327-346, Flow.sample_and_log_prob; the proposal draw of mcmc.py:49-110) at more than 32
// dimensions: z ~ N(0, I) from the counter-based generator exactly as the d <= 32 kernels draw it (fp32 Box-Muller quads keyed
// by the global particle index: coordinate j is element j % 4 of quad j / 4), the layers inverted in reverse order on the same
// layer code, x = x' scale + loc and log q(x) = N(z) - sum s - sum log scale from the same pass.  A coupling layer inverts in
// one evaluation; an autoregressive transform by fixed-point passes x <- z exp(s(x)) + t(x) from x = 0 (asmc_flow.hip,
// k_maf_sample: at most d passes, stopping at the first pass that returns its input bit for bit - here for every particle of
// the BLOCK, whose waves share the weight stream; the operand range is checked over all passes).
template <int KIND, int D, int W, typename XT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_flow16_sample(int64_t n, int d, const float* __restrict__ packed, int n_layers,
                                                          const float* __restrict__ loc, const float* __restrict__ scale, float ladj0,
                                                          float base_const, unsigned long long seed, unsigned long long gid0,
                                                          uint32_t draw_id, XT* __restrict__ x, double* __restrict__ out, int all_passes, int form) {
    using FD = Flow16<KIND, D, W>;
    extern __shared__ __align__(16) float sm[];
    constexpr int WAVES = THREADS / 64, SL = FD::SL, CS = FD::CS;
    float* slots = sm;
    float* s_bias = slots + 2 * FD::CW;
    Flow16Stream<FD, THREADS> stream;
    stream.start(packed + (size_t)n_layers * FD::BIAS, slots, n_layers, FD::MAF ? F16_REPEAT : F16_BACKWARD);
    for (int e = threadIdx.x; e < n_layers * FD::BIAS; e += THREADS) s_bias[e] = packed[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, h = lane >> 4;
    const int64_t n_groups = (n + 15) / 16;
    const int64_t per_round = (int64_t)gridDim.x * WAVES;
    const int64_t rounds = (n_groups + per_round - 1) / per_round;
    for (int64_t it = 0; it < rounds; it++) {
        const int64_t g = it * per_round + (int64_t)blockIdx.x * WAVES + wave;
        const int64_t row = g * 16 + p;
        const bool valid = row < n;
        const unsigned long long gid = gid0 + (unsigned long long)(valid ? row : (n - 1));
        F16State<FD> z;
        float q = 0.0f;
#pragma unroll
        for (int sp = 0; sp < SL / 2; sp++) {  // the lane's coordinate pairs: two elements of one quad (the pair starts at an even coordinate)
            const int j0 = f16_nat(KIND, D, d, f16_coord(2 * sp, h)), j1 = f16_nat(KIND, D, d, f16_coord(2 * sp + 1, h));
            double zq[4] = {0.0, 0.0, 0.0, 0.0};
            if (j0 >= 0) normal_quad_f32(seed, gid, draw_id, (uint32_t)(j0 >> 2), zq[0], zq[1], zq[2], zq[3]);
            float z0 = 0.0f, z1 = 0.0f;
            if (j0 >= 0) z0 = (float)((j0 & 3) == 0 ? zq[0] : (j0 & 3) == 1 ? zq[1] : (j0 & 3) == 2 ? zq[2] : zq[3]);
            if (j1 >= 0) {
                if ((j1 >> 2) != (j0 >> 2)) normal_quad_f32(seed, gid, draw_id, (uint32_t)(j1 >> 2), zq[0], zq[1], zq[2], zq[3]);
                z1 = (float)((j1 & 3) == 0 ? zq[0] : (j1 & 3) == 1 ? zq[1] : (j1 & 3) == 2 ? zq[2] : zq[3]);
            }
            z.set(2 * sp, z0);
            z.set(2 * sp + 1, z1);
            q = fmaf(z0, z0, fmaf(z1, z1, q));
        }
        float ladj = 0.0f, amax = 0.0f;
        if constexpr (FD::MAF) {
            for (int c = n_layers - 1; c >= 0; c--) {
                float xv[CS];
#pragma unroll
                for (int i = 0; i < CS; i++) xv[i] = 0.0f;
                float ladj_pass = 0.0f;
                for (int pass = 0; pass < d; pass++) {
                    float tr[CS];
#pragma unroll
                    for (int i = 0; i < CS; i++) tr[i] = z.a[i];
                    ladj_pass = 0.0f;
                    unsigned amax_pk = 0u;
                    f16_layer<FD, W, THREADS, true>(xv, tr, s_bias + c * FD::BIAS, stream, lane, ladj_pass, amax_pk, form);
                    bool same = true, nan = false;
#pragma unroll
                    for (int i = 0; i < CS; i++) {
                        same = same && (__float_as_uint(tr[i]) == __float_as_uint(xv[i]));
                        nan = nan || (tr[i] != tr[i]);
                    }
                    float am = range_pk_max(amax_pk);
                    am = (am != am || nan) ? __builtin_inff() : am;
                    amax = fmaxf(amax, am);  // every pass counts (asmc_flow.hip, k_maf_sample)
#pragma unroll
                    for (int i = 0; i < CS; i++) xv[i] = pass + 1 < d ? fminf(fmaxf(tr[i], -60000.0f), 60000.0f) : tr[i];
                    // the fixed point, for every particle of the block (its waves share the stream: a block-uniform decision)
                    if (!all_passes && __syncthreads_and(same ? 1 : 0)) break;
                }
                ladj += ladj_pass;
#pragma unroll
                for (int i = 0; i < CS; i++) z.a[i] = xv[i];
                // the stream has put this transform's first chunk into flight again; the next one needed is the transform below
                // (or, after the last, the top transform of the next round)
                stream.redirect(c > 0 ? c - 1 : n_layers - 1);
            }
        } else {
            unsigned amax_pk = 0u;
            for (int c = n_layers - 1; c >= 0; c--) {
                if ((c & 1) == 0)
                    f16_layer<FD, W, THREADS, true>(z.a, z.b, s_bias + c * FD::BIAS, stream, lane, ladj, amax_pk);
                else
                    f16_layer<FD, W, THREADS, true>(z.b, z.a, s_bias + c * FD::BIAS, stream, lane, ladj, amax_pk);
                __builtin_amdgcn_sched_barrier(0);
            }
            amax = range_pk_max(amax_pk);
            amax = (amax != amax) ? __builtin_inff() : amax;
        }
        q = f16_quad_sum(q);
        const float lj = f16_quad_sum(ladj);
        const float am = f16_quad_max(amax);
        const float val = !(am < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
        if (valid) {
#pragma unroll
            for (int s = 0; s < SL; s++) {
                const int j = f16_nat(KIND, D, d, f16_coord(s, h));
                if (j >= 0) x[row * d + j] = (XT)(z.get(s) * scale[j] + loc[j]);
            }
            if (h == 0) out[row] = (double)val;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int KIND, int D, int W, typename XT>
static int launch_flow16_sample(asmc_ctx* ctx, int64_t n, const asmc_coupling* f, unsigned long long seed, unsigned long long gid0,
                                uint32_t draw_id, XT* x, double* out, hipStream_t st) {
    using FD = Flow16<KIND, D, W>;
    constexpr int THREADS = 512;
    const size_t lds = (size_t)(2 * FLOW16_CHUNK_WORDS + f->n_layers * FD::BIAS) * sizeof(float);
    ASMC_REQUIRE(lds <= 160 * 1024, "flow16: biases exceed the LDS");
    auto kern = k_flow16_sample<KIND, D, W, XT, THREADS>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t n_groups = (n + 15) / 16;
    const int64_t want = (n_groups + THREADS / 64 - 1) / (THREADS / 64);
    const int grid = (int)(want < (int64_t)ctx->num_cu * 2 ? want : (int64_t)ctx->num_cu * 2);
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    ASMC_LAUNCH(ctx, st, "k_flow16_sample", kern, dim3(grid), dim3(THREADS), lds, st, n, (int)f->dims, f->packed_dev, (int)f->n_layers,
                f->loc_dev, f->scale_dev, ladj0, base_const, seed, gid0, draw_id, x, out, getenv("ASMC_MAF_SAMPLE_ALL_PASSES") ? 1 : 0, (int)f->affine);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_flow16_sample(asmc_ctx* ctx, int64_t n, int x_dtype, const asmc_coupling* f, unsigned long long seed, unsigned long long gid0,
                       uint32_t draw_id, void* x_out, double* lq_out, hipStream_t st) {
    if (!asmc_flow_math_split()) {
        asmc_set_error("flows of more than 32 dimensions run the split-fp16 layers only (ASMC_FLOW_MATH=f32 is not available there)");
        return ASMC_ERR_UNSUPPORTED;
    }
    const int D = f16_pad_dim(f->dims);
#define X(K, DD, WW)                                                                                                                     \
    if (f->kind == K && D == DD && f->hidden == WW) {                                                                                    \
        if (x_dtype == ASMC_F64) return launch_flow16_sample<K, DD, WW, double>(ctx, n, f, seed, gid0, draw_id, (double*)x_out, lq_out, st); \
        return launch_flow16_sample<K, DD, WW, float>(ctx, n, f, seed, gid0, draw_id, (float*)x_out, lq_out, st);                        \
    }
    F16_SHAPES(X)
#undef X
    asmc_set_error("flow16 sample: unsupported shape (kind %d, dims %d, hidden %d)", (int)f->kind, (int)f->dims, (int)f->hidden);
    return ASMC_ERR_UNSUPPORTED;
}
