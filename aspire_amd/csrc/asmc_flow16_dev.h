// asmc_flow16_dev.h — neural-flow densities for 32 < d <= 128 on 16-particle groups (round 5).
//
// The d <= 32 kernels (asmc_flow_dev.h) give a wave two 32-particle tiles with the lane halves splitting a particle's
// coordinates, and keep every layer of the flow resident in LDS.  Neither survives d = 64 / 128: four coupling layers of a
// 64-dimensional flow are 164 KB of split-fp16 operands, and a lane half would hold 32 - 64 coordinates next to 64
// accumulator registers.  Here a wave owns SIXTEEN particles and a particle is spread over FOUR lanes - the layout of the
// d = 64 / 128 pCN kernels (asmc_pcn_mm.hip): lane (p = l & 15, h = l >> 4) holds the D / 4 coordinates
//     coordinate(s, h) = 8 (s / 2) + 2 h + s % 2,   s = 0 .. D / 4 - 1                               (mm_coord)
// of particle p, so the proposal x' = mu + L y' (fp64 MFMA 16x16x4, N = 16 particles) hands its result to the flow without a
// shuffle.  The dense layers run on v_mfma_f32_16x16x32_f16 (A = weights [16 units x 32 inputs], B = activations [32 inputs
// x 16 particles], fp32 accumulation) as the same three split-fp16 products per K step as the d <= 32 kernels
// (asmc_flow_dev.h: hi hi, hi lo, lo hi).  In the 16 x 16 accumulator lane (p, g) holds rows 4 g + r of column p: with the
// contraction order of the next layer chosen as   k(S, g, j) = 16 (2 S + j / 4) + 4 g + j % 4   (K step S, lane group g, slot
// j of the lane's eight) the accumulators of one layer ARE the B operand of the next - activations never leave the lane, and
// the first layer takes its inputs straight from the lane's own coordinates (the permutations live in the packed weights).
// Output rows are packed so that (s_raw, t) of a coordinate land in the lane that holds it, s_raw and t blocks alternating
// (block 2 m: s_raw of the lane's slots 4 m .. 4 m + 3, block 2 m + 1: their t) - the epilogue runs per pair of blocks and
// the output accumulators need eight registers however wide the flow is.
//
// Weights are STREAMED: the packed operand images (fp16 pairs, split on the host) are cut into chunks of whole output
// blocks, at most FLOW16_CHUNK_WORDS words each; two LDS slots take them in turn (global_load_lds, no staging registers),
// chunk q + 1 in flight while the block's waves compute on chunk q, one workgroup barrier per chunk.  Every wave of a block
// therefore walks the flow in lockstep, one 16-particle group per wave and round.
// Coupling flows (flows.py CouplingFlow; RealNVP) and masked autoregressive flows (MAFFlow; the reference's default class,
// flows/torch/flows.py:140-168) share the code: a coupling layer conditions on the lane's first (or second) D / 8 slots and
// transforms the others, an autoregressive transform conditions on and transforms all D / 4.
#pragma once
#include "asmc_common.h"
#include "asmc_flow_dev.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define FLOW16_CHUNK_WORDS 8192  // default: 32 KB per LDS slot

__host__ __device__ constexpr int f16_coord(int s, int h) { return 8 * (s / 2) + 2 * h + (s % 2); }  // = mm_coord (asmc_pcn_mm.hip)

// CW: words per LDS slot of the weight stream.  A slot that takes a whole layer's operand images makes the layer ONE chunk
// (WHOLE: one barrier per layer); otherwise a chunk is a run of whole output blocks of one matrix.
template <int KIND, int D, int W, int CWV = FLOW16_CHUNK_WORDS>
struct Flow16 {
    static constexpr int CW = CWV;
    static_assert(D == 64 || D == 128, "padded dimension");
    static_assert(W == 32 || W == 64 || W == 128, "hidden width");
    static constexpr bool MAF = KIND == ASMC_FLOW_MAF;
    static constexpr int SL = D / 4;                  // coordinate slots per lane
    static constexpr int CS = MAF ? SL : SL / 2;      // conditioner slots per lane = transformed slots per lane
    static constexpr int KS1 = CS / 8;                // K = 32 steps of the first dense layer
    static constexpr int KS2 = W / 32;                // ... of a layer that reads a hidden layer
    static constexpr int NB1 = W / 16;                // 16-row output blocks of a hidden layer
    static constexpr int NB3 = CS / 2;                // ... of the output layer: (s_raw, t) x CS slots x 4 lanes / 16
    static constexpr int OUT = NB3 * 16;
    static constexpr int BIAS = 2 * W + OUT;          // floats per layer (b1 | b2 | b3 in lane order)
    static constexpr int BLK1 = KS1 * 512, BLK2 = KS2 * 512;  // 4-byte words per output block: (hi, lo) images of every K step
    static constexpr int A1 = NB1 * BLK1, A2 = NB1 * BLK2, A3 = NB3 * BLK2;
    static constexpr int LAYER_A = A1 + A2 + A3;      // words of operand images per layer
    static constexpr bool WHOLE = CW >= LAYER_A;
    // blocks per chunk (whole blocks, <= CW words) and chunks per matrix
    static constexpr int BC1 = CW / BLK1 < NB1 ? CW / BLK1 : NB1;
    static constexpr int BC2 = CW / BLK2 < NB1 ? CW / BLK2 : NB1;
    static constexpr int BC3r = CW / BLK2 < NB3 ? CW / BLK2 : NB3;
    static constexpr int BC3 = BC3r < NB3 ? (BC3r / 2) * 2 : BC3r;  // (s_raw, t) block pairs stay in one chunk
    static_assert(BC1 >= 1 && BC2 >= 1 && BC3 >= 2, "an output block pair must fit a chunk");
    static_assert(NB1 % BC1 == 0 && NB1 % BC2 == 0 && NB3 % BC3 == 0, "chunks of equal size");
    static constexpr int P1 = NB1 / BC1, P2 = NB1 / BC2, P3 = NB3 / BC3;
    static constexpr int NPARTS = WHOLE ? 1 : P1 + P2 + P3;  // chunks per layer
    // word offset and length of chunk i of a layer (relative to the layer's operand images)
    static constexpr __host__ __device__ int part_off(int i) {
        return WHOLE ? 0 : i < P1 ? i * BC1 * BLK1 : i < P1 + P2 ? A1 + (i - P1) * BC2 * BLK2 : A1 + A2 + (i - P1 - P2) * BC3 * BLK2;
    }
    static constexpr __host__ __device__ int part_words(int i) {
        return WHOLE ? LAYER_A : i < P1 ? BC1 * BLK1 : i < P1 + P2 ? BC2 * BLK2 : BC3 * BLK2;
    }
};

// words of the packed block: [biases of every layer][operand images of every layer]
template <int KIND, int D, int W>
__host__ __device__ constexpr int64_t flow16_words(int n_layers) {
    return (int64_t)n_layers * (Flow16<KIND, D, W>::BIAS + Flow16<KIND, D, W>::LAYER_A);
}

// eight fp32 values -> their (hi, lo) fp16 operand halves; the range check rides on the hi halves (asmc_flow_dev.h)
template <bool RELU, bool PROP = false>  // (PROP: asmc_flow_dev.h split2_f16 - NaN-propagating ReLU, no range check)
__device__ __forceinline__ void f16_split8(const float (&x)[8], half8& hi, half8& lo, unsigned& amax_pk) {
    unsigned hp[4], lp[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (c == 3)
            split2_f16<RELU, true, PROP>(x[2 * c], x[2 * c + 1], hp[c], lp[c]);
        else
            split2_f16<RELU, false, PROP>(x[2 * c], x[2 * c + 1], hp[c], lp[c]);
    }
    if (!PROP) {
        split4_range<!RELU>(hp[0], hp[1], amax_pk);
        split4_range<!RELU>(hp[2], hp[3], amax_pk);
    }
    hi = __builtin_bit_cast(half8, flow_u4{hp[0], hp[1], hp[2], hp[3]});
    lo = __builtin_bit_cast(half8, flow_u4{lp[0], lp[1], lp[2], lp[3]});
}

// The stream of weight chunks through two LDS slots.  `slots`: 2 x FD::CW words of LDS; `gA`: the flow's
// operand images in HBM (word 0 = chunk 0 of layer 0).  next() closes the previous chunk (every wave of the block is
// through with it), makes the current one visible and puts the one after it into flight; it returns the current chunk.
// Which layer follows a layer's last chunk is the stream's `mode`: F16_FORWARD (the density: 0, 1, .., n - 1, then 0 again for
// the next round), F16_BACKWARD (sampling a coupling flow: n - 1, .., 0, then n - 1) or F16_REPEAT (the passes that invert one
// autoregressive transform: the same layer again; redirect() replaces a chunk that was put into flight on that assumption).
#define F16_FORWARD 0
#define F16_BACKWARD 1
#define F16_REPEAT 2
template <class FD, int THREADS>
struct Flow16Stream {
    const float* __restrict__ gA;
    float* slots;
    int n_layers, mode;
    int layer, part;  // the chunk that is IN FLIGHT (issued, not yet waited for)
    unsigned parity;  // slot it goes to

    __device__ __forceinline__ void issue(int l, int p, unsigned slot) {
        const int off = l * FD::LAYER_A + FD::part_off(p);
        const int words = FD::part_words(p);
        // wave w takes the 1 KiB pieces w, w + WAVES, ... of the chunk (the LDS side of a piece is wave-uniform base + lane * 16
        // bytes: contiguous in exactly the order of the source)
        constexpr int WAVES = THREADS / 64;
        // (the wave index as a SCALAR: derived from threadIdx.x it counts as divergent, and the LDS base of global_load_lds - the M0
        // register - then sits in a waterfall loop that the compiler wraps around each chunk's whole block of matrix instructions)
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
        for (int q = wave; q * 256 < words; q += WAVES) {
            const float* src = gA + off + q * 256 + lane * 4;
            float* dst = slots + slot * FD::CW + q * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
    // called once, before the first round: the first chunk of layer `l0` into slot 0
    __device__ __forceinline__ void start(const float* g, float* s, int nl, int md = F16_FORWARD) {
        gA = g, slots = s, n_layers = nl, mode = md;
        layer = md == F16_FORWARD ? 0 : nl - 1, part = 0, parity = 0;
        issue(layer, 0, 0);
    }
    __device__ __forceinline__ const float* next() {
        // the chunk in flight has landed (this wave's share: vmcnt; everybody's: the barrier), and every wave is past its reads
        // of the chunk before it - whose slot the chunk after this one may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* cur = slots + parity * FD::CW;
        int l = layer, p = part + 1;
        if (p == FD::NPARTS) {
            p = 0;
            if (mode == F16_FORWARD) l = (l + 1 == n_layers) ? 0 : l + 1;
            if (mode == F16_BACKWARD) l = (l == 0) ? n_layers - 1 : l - 1;
        }
        layer = l, part = p, parity ^= 1u;
        issue(l, p, parity);
        return cur;
    }
    // the chunk in flight is not the one the caller needs next (block-uniform decision): chunk 0 of layer l instead
    __device__ __forceinline__ void redirect(int l) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        layer = l, part = 0;
        issue(l, 0, parity);
    }
};

// acc[b] (b = 0 .. NBC - 1: the chunk's output blocks) = bias + A_b * B over KS K = 32 steps, three fp16 products per step.
// A: LDS image of the chunk ([block][K step][hi | lo][lane] x 16 bytes); bias: the layer's bias row in lane order (float4 per
// (block, lane group)), already offset to the chunk's first block.
template <int NBC, int KS>
__device__ __forceinline__ void f16_dense(floatx4 (&acc)[NBC], const half8 (&bh)[KS], const half8 (&bl)[KS], const float* __restrict__ A,
                                          const float* __restrict__ bias, int lane) {
    const half8* Ap = reinterpret_cast<const half8*>(A) + lane;
    const int g = lane >> 4;
#pragma unroll
    for (int b = 0; b < NBC; b++) acc[b] = *reinterpret_cast<const floatx4*>(bias + (b * 4 + g) * 4);
#pragma unroll
    for (int b0 = 0; b0 < NBC; b0 += 2) {
        constexpr int PB = 2;
#pragma unroll
        for (int S = 0; S < KS; S++) {
            half8 ah[PB], al[PB];
#pragma unroll
            for (int e = 0; e < PB; e++) {
                if (b0 + e < NBC) {
                    ah[e] = Ap[(size_t)(((b0 + e) * KS + S) * 2) * 64];
                    al[e] = Ap[(size_t)(((b0 + e) * KS + S) * 2 + 1) * 64];
                }
            }
#pragma unroll
            for (int e = 0; e < PB; e++)
                if (b0 + e < NBC) acc[b0 + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[e], bh[S], acc[b0 + e], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < PB; e++)
                if (b0 + e < NBC) acc[b0 + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[e], bl[S], acc[b0 + e], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < PB; e++)
                if (b0 + e < NBC) acc[b0 + e] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[e], bh[S], acc[b0 + e], 0, 0, 0);
        }
    }
}

// One layer (coupling layer or autoregressive transform) of one 16-particle group.  cond: the lane's CS conditioner inputs,
// trans: its CS transformed coordinates (updated in place).  A coupling layer passes its two halves (which is which alternates
// with the layer); an autoregressive transform passes the same array twice for the density (every conditioner input is
// converted before the first coordinate changes) and (current estimate, latent copy) for a pass of its inversion.
// INVERSE: the sampling direction x_b = z_b exp(s) + t.
// FORM: the affine form (asmc_flow_dev.h flow_affine) as a compile-time constant, or -1: the run-time argument `form` (a uniform
// branch per coordinate then sits between the output layer's matrix instructions and cuts the scheduler's blocks: the density
// kernels pass the constant).
template <class FD, int W, int THREADS, bool INVERSE = false, int FORM = -1, bool PROP = false>
__device__ __forceinline__ void f16_layer(const float (&cond)[FD::CS], float (&trans)[FD::CS], const float* __restrict__ bias,
                                          Flow16Stream<FD, THREADS>& stream, int lane, float& ladj, unsigned& amax_pk, int form = 0) {
    constexpr int KS1 = FD::KS1, KS2 = FD::KS2, NB1 = FD::NB1;
    // ---- first dense layer: conditioner slots -> hidden 1
    // (The operand is converted IN FRONT of the stream's barrier, never right in front of the products: the lo halves are written by
    // inline assembly (split2_f16), which the compiler's hazard recogniser does not see - placed behind the barrier, the register
    // allocator reused a first product's bias accumulator (SrcC, still being read by the matrix pipe) as their destination and the
    // first layer came out wrong; round 6, found by tools/debug_flow16_step_lq.py.)
    half8 bh1[KS1], bl1[KS1];
#pragma unroll
    for (int S = 0; S < KS1; S++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = cond[8 * S + j];
        f16_split8<false, PROP>(v, bh1[S], bl1[S], amax_pk);
    }
    floatx4 h1[NB1];
    const float* Aw = nullptr;  // WHOLE: the layer's one chunk
    if constexpr (FD::WHOLE) Aw = stream.next();
#pragma unroll
    for (int p = 0; p < FD::P1; p++) {
        const float* A = FD::WHOLE ? Aw + (size_t)p * FD::BC1 * FD::BLK1 : stream.next();
        floatx4 part[FD::BC1];
        f16_dense<FD::BC1, KS1>(part, bh1, bl1, A, bias + p * FD::BC1 * 16, lane);
#pragma unroll
        for (int b = 0; b < FD::BC1; b++) h1[p * FD::BC1 + b] = part[b];
    }
    // ---- second: relu(hidden 1) -> hidden 2.  K step S takes the lane's registers of blocks 2 S, 2 S + 1
    half8 bh2[KS2], bl2[KS2];
#pragma unroll
    for (int S = 0; S < KS2; S++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = h1[2 * S + j / 4][j % 4];
        f16_split8<true, PROP>(v, bh2[S], bl2[S], amax_pk);
    }
    floatx4 h2[NB1];
#pragma unroll
    for (int p = 0; p < FD::P2; p++) {
        const float* A = FD::WHOLE ? Aw + FD::A1 + (size_t)p * FD::BC2 * FD::BLK2 : stream.next();
        floatx4 part[FD::BC2];
        f16_dense<FD::BC2, KS2>(part, bh2, bl2, A, bias + W + p * FD::BC2 * 16, lane);
#pragma unroll
        for (int b = 0; b < FD::BC2; b++) h2[p * FD::BC2 + b] = part[b];
    }
    // ---- output layer, a pair of blocks at a time: (s_raw, t) of four of the lane's transformed slots
#pragma unroll
    for (int S = 0; S < KS2; S++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = h2[2 * S + j / 4][j % 4];
        f16_split8<true, PROP>(v, bh2[S], bl2[S], amax_pk);
    }
    // The epilogue of a pair (exp / rcp chains: vector and transcendental work) is written BEHIND the matrix instructions of the next
    // pair, which it does not depend on - one basic block, so the scheduler places it between them (round 6; round 5 ran every
    // pair's epilogue between two bursts of matrix instructions: 12 MFMAs, ~90 vector instructions with their wait states, ...).
    floatx4 po[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    flow_f2 lacc = {0.0f, 0.0f};  // the layer's log-determinant per coordinate parity (flow_affine2), folded into ladj at the end
    auto epilogue = [&](const floatx4 (&o)[2], int m) {  // the lane's transformed slots 4 m .. 4 m + 3, two at a time
#pragma unroll
        for (int r = 0; r < 4; r += 2)  // (asmc_flow_dev.h: form 0 = 2 tanh(sraw / 2); 1 = zuko's soft clip)
            flow_affine2<INVERSE>(trans[4 * m + r], trans[4 * m + r + 1], o[0][r], o[0][r + 1], o[1][r], o[1][r + 1], lacc, FORM >= 0 ? FORM : form);
    };
#pragma unroll
    for (int p = 0; p < FD::P3; p++) {
        const float* A = FD::WHOLE ? Aw + FD::A1 + FD::A2 + (size_t)p * FD::BC3 * FD::BLK2 : stream.next();
#pragma unroll
        for (int pr = 0; pr < FD::BC3 / 2; pr++) {
            floatx4 o[2];
            f16_dense<2, KS2>(o, bh2, bl2, A + (size_t)pr * 2 * FD::BLK2, bias + 2 * W + (p * FD::BC3 + 2 * pr) * 16, lane);
            const int m = p * (FD::BC3 / 2) + pr;
            if (m > 0) epilogue(po, m - 1);
            po[0] = o[0], po[1] = o[1];
        }
    }
    epilogue(po, FD::P3 * (FD::BC3 / 2) - 1);
    ladj += lacc.x + lacc.y;
}

// the lane's standardised coordinates: slot s of the lane is a[s] (autoregressive flows: every slot; coupling flows: the
// first-half slots s < CS) or b[s - CS] (coupling flows: the second-half slots)
template <class FD>
struct F16State {
    float a[FD::CS];
    float b[FD::MAF ? 1 : FD::CS];
    __device__ __forceinline__ void set(int s, float v) {  // (s: compile-time constant after unrolling)
        if (FD::MAF || s < FD::CS) a[s] = v;
        else b[FD::MAF ? 0 : s - FD::CS] = v;
    }
    __device__ __forceinline__ float get(int s) const { return (FD::MAF || s < FD::CS) ? a[s] : b[FD::MAF ? 0 : s - FD::CS]; }
};

// sum over the four lanes p, p + 16, p + 32, p + 48 of a particle
__device__ __forceinline__ float f16_quad_sum(float q) {
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    return q;
}
__device__ __forceinline__ float f16_quad_max(float q) {
    q = fmaxf(q, __shfl_xor(q, 16, 64));
    q = fmaxf(q, __shfl_xor(q, 32, 64));
    return q;
}

// log q of the group's particles from the lane's standardised coordinates (every lane of a particle returns it)
template <class FD, int W, int THREADS, int FORM = -1, bool PROP = false>
__device__ __forceinline__ float f16_logprob(F16State<FD>& x, int n_layers, const float* __restrict__ biases,
                                             Flow16Stream<FD, THREADS>& stream, int lane, float ladj0, float base_const, int form = 0) {
    float ladj = 0.0f;
    unsigned amax_pk = 0u;
    for (int c = 0; c < n_layers; c += 2) {  // (n_layers is uniform over the block: every wave meets the same barriers)
        if constexpr (FD::MAF) {
            f16_layer<FD, W, THREADS, false, FORM, PROP>(x.a, x.a, biases + c * FD::BIAS, stream, lane, ladj, amax_pk, form);
        } else {
            f16_layer<FD, W, THREADS, false, 0, PROP>(x.a, x.b, biases + c * FD::BIAS, stream, lane, ladj, amax_pk);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < n_layers) {
            if constexpr (FD::MAF) {
                f16_layer<FD, W, THREADS, false, FORM, PROP>(x.a, x.a, biases + (c + 1) * FD::BIAS, stream, lane, ladj, amax_pk, form);
            } else {
                f16_layer<FD, W, THREADS, false, 0, PROP>(x.b, x.a, biases + (c + 1) * FD::BIAS, stream, lane, ladj, amax_pk);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float q = 0.0f;
#pragma unroll
    for (int s = 0; s < FD::SL; s++) q = fmaf(x.get(s), x.get(s), q);
    q = f16_quad_sum(q);
    const float lj = f16_quad_sum(ladj);
    float am = range_pk_max(amax_pk);
    am = (am != am) ? __builtin_inff() : am;
    am = f16_quad_max(am);
    return !(am < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
}
