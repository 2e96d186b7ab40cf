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
#include <stdlib.h>

#include "asmc_common.h"

#include "asmc_flow_dev.h"
#include "asmc_pcn_dev.h"  // normal_quad_f32 (the sampling kernel draws its latent from the counter-based generator)

template <int H, int W, typename XT, int FLOW_THREADS, int TPW, bool HS>
__global__ __launch_bounds__(FLOW_THREADS) void k_coupling_logprob(int64_t n, int d, const XT* __restrict__ x,
                                                                  const float* __restrict__ packed, int n_layers,
                                                                  int resident, const float* __restrict__ loc,
                                                                  const float* __restrict__ scale, float ladj0,
                                                                  float base_const, double* __restrict__ out) {
    extern __shared__ __align__(16) float sp[];
    using FD = FlowDims<H, W>;
    constexpr int FLOW_WAVES = FLOW_THREADS / 64;
    constexpr int ROWS = 32 * TPW;  // particles per wave and round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, hh = lane >> 5;
    const int dh = d / 2;
    if (resident && HS) {
        flow_stage_hs<H, W, FLOW_THREADS>(sp, packed, n_layers);  // split-fp16 operand images (asmc_flow_dev.h)
        __syncthreads();
    } else if (resident) {
        // all of a thread's loads are issued before the first LDS store: 14 dependent round trips to L2 would
        // otherwise cost ~5 % of the kernel at 1M particles
        const int total4 = n_layers * FD::LAYER / 4;
        for (int base = 0; base < total4; base += FLOW_THREADS * 8) {
            float4 tmp[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i4 = base + q * FLOW_THREADS + threadIdx.x;
                tmp[q] = i4 < total4 ? reinterpret_cast<const float4*>(packed)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i4 = base + q * FLOW_THREADS + threadIdx.x;
                if (i4 < total4) reinterpret_cast<float4*>(sp)[i4] = tmp[q];
            }
        }
        __syncthreads();
    }
    const int64_t n_tiles = (n + ROWS - 1) / ROWS;
    const int64_t tiles_per_round = (int64_t)gridDim.x * FLOW_WAVES;
    const int64_t rounds = (n_tiles + tiles_per_round - 1) / tiles_per_round;  // same trip count for every wave
    // raw rows of the NEXT tile are fetched while the current one is in the matrix pipe (TPW = 1: the registers
    // are there; with two tiles per wave the accumulators leave no room and the loads stay at the loop head)
    constexpr bool PREFETCH = (TPW == 1);
    XT ra[TPW][H / 2], rb[TPW][H / 2];
    auto load_raw = [&](int64_t tile) {
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
            const int64_t row = tile * ROWS + tt * 32 + p;
            const bool valid = tile < n_tiles && row < n;
#pragma unroll
            for (int i = 0; i < H / 2; i++) ra[tt][i] = rb[tt][i] = (XT)0;
            if (!valid) continue;
            if (d == 2 * H) {  // unpadded rows: each lane's two segments as 16-byte vector loads
                constexpr int V = 16 / (int)sizeof(XT);
                struct alignas(16) Vec {
                    XT e[V];
                };
                const Vec* pa = reinterpret_cast<const Vec*>(x + row * d + hh * (H / 2));
                const Vec* pb = reinterpret_cast<const Vec*>(x + row * d + H + hh * (H / 2));
#pragma unroll
                for (int q = 0; q < (H / 2) / V; q++) {
                    const Vec va = pa[q], vb = pb[q];
#pragma unroll
                    for (int e = 0; e < V; e++) {
                        ra[tt][q * V + e] = va.e[e];
                        rb[tt][q * V + e] = vb.e[e];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < H / 2; i++) {
                    const int jp = hh * (H / 2) + i;
                    if (jp < dh) {
                        ra[tt][i] = x[row * d + jp];
                        rb[tt][i] = x[row * d + dh + jp];
                    }
                }
            }
        }
    };
    if (PREFETCH) load_raw((int64_t)blockIdx.x * FLOW_WAVES + wave);
    for (int64_t it = 0; it < rounds; it++) {
        const int64_t tile = (it * gridDim.x + blockIdx.x) * FLOW_WAVES + wave;
        if (!PREFETCH) load_raw(tile);
        float xa[TPW][H / 2], xb[TPW][H / 2];
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
#pragma unroll
            for (int i = 0; i < H / 2; i++) {
                const int jp = hh * (H / 2) + i;  // padded dims: loc = 0, scale = 1 are not stored, keep them at zero
                const bool real = jp < dh;
                xa[tt][i] = real ? flow_standardise((float)ra[tt][i], loc[jp], scale[jp], 1.0f / scale[jp]) : 0.0f;
                xb[tt][i] = real ? flow_standardise((float)rb[tt][i], loc[dh + jp], scale[dh + jp], 1.0f / scale[dh + jp]) : 0.0f;
            }
        }
        if (PREFETCH && it + 1 < rounds) load_raw(((it + 1) * gridDim.x + blockIdx.x) * FLOW_WAVES + wave);
        float ladj[TPW];
        float amax = 0.0f;  // largest operand the split-fp16 layers converted (this lane's particle; TPW = 1 there)
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) ladj[tt] = 0.0f;
        for (int c = 0; c < n_layers; c++) {
            const float* lp = sp + (resident ? (size_t)c * FD::LAYER : 0);
            if (!resident) {
                __syncthreads();  // every wave is done with the previous layer's block
                if (HS)
                    flow_stage_hs<H, W, FLOW_THREADS>(sp, packed + (size_t)c * FD::LAYER, 1);
                else
                    for (int i = threadIdx.x * 4; i < FD::LAYER; i += FLOW_THREADS * 4)
                        *reinterpret_cast<float4*>(sp + i) =
                            *reinterpret_cast<const float4*>(packed + (size_t)c * FD::LAYER + i);
                __syncthreads();
            }
            if constexpr (HS) {
                static_assert(TPW == 1, "the split-fp16 layers take one tile per wave");
                if ((c & 1) == 0)
                    coupling_layer_hs<H, W>(xa[0], xb[0], lp, lane, hh, ladj[0], amax);
                else
                    coupling_layer_hs<H, W>(xb[0], xa[0], lp, lane, hh, ladj[0], amax);
            } else if ((c & 1) == 0)
                coupling_layer<H, W, TPW>(xa, xb, lp, lane, hh, ladj);
            else
                coupling_layer<H, W, TPW>(xb, xa, lp, lane, hh, ladj);
        }
#pragma unroll
        for (int tt = 0; tt < TPW; tt++) {
            const int64_t row = tile * ROWS + tt * 32 + p;
            float q = 0.0f;
#pragma unroll
            for (int i = 0; i < H / 2; i++) q += xa[tt][i] * xa[tt][i] + xb[tt][i] * xb[tt][i];
            q += __shfl_xor(q, 32);
            const float lj = ladj[tt] + __shfl_xor(ladj[tt], 32);
            const float am = fmaxf(amax, __shfl_xor(amax, 32));
            // an operand past the fp16 range makes the split products garbage: report NaN, not a finite wrong number
            const float val = (HS && !(am < FLOW_HS_MAX)) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
            if (tile < n_tiles && row < n && hh == 0) out[row] = (double)val;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sampling direction (reference flows/torch/flows.py:327-346, Flow.sample_and_log_prob): z ~ N(0, I) from the counter-based
// generator (fp32 Box-Muller quads of asmc_pcn_dev.h, keyed by the global particle index: the draw does not depend on how
// the population is sharded), the coupling layers inverted in reverse order on the same split-fp16 conditioner code,
// x = z' scale + loc, and log q(x) = N(z) - sum s - sum log scale from the same pass.  Same tile layout as the density kernel.
template <int H, int W, typename XT, int FLOW_THREADS>
__global__ __launch_bounds__(FLOW_THREADS) void k_coupling_sample(int64_t n, int d, const float* __restrict__ packed, int n_layers,
                                                                 const float* __restrict__ loc, const float* __restrict__ scale,
                                                                 float ladj0, float base_const, unsigned long long seed,
                                                                 unsigned long long gid0, uint32_t draw_id, XT* __restrict__ x,
                                                                 double* __restrict__ out) {
    extern __shared__ __align__(16) float sp[];
    using FD = FlowDims<H, W>;
    constexpr int FLOW_WAVES = FLOW_THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, hh = lane >> 5;
    const int dh = d / 2;
    flow_stage_hs<H, W, FLOW_THREADS>(sp, packed, n_layers);
    __syncthreads();
    const int64_t n_tiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * FLOW_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * FLOW_WAVES) {
        const int64_t row = tile * 32 + p;
        const unsigned long long gid = gid0 + (unsigned long long)(row < n ? row : n - 1);
        float xa[H / 2], xb[H / 2];
        float q = 0.0f;
        {   // coordinate c is element c % 4 of quad c / 4 of this particle's draw
            int have = -1;
            double zq[4] = {0.0, 0.0, 0.0, 0.0};
            auto normal_at = [&](int c) {
                if ((c >> 2) != have) {
                    have = c >> 2;
                    normal_quad_f32(seed, gid, draw_id, (uint32_t)have, zq[0], zq[1], zq[2], zq[3]);
                }
                return (float)zq[c & 3];
            };
#pragma unroll
            for (int i = 0; i < H / 2; i++) {
                const int jp = hh * (H / 2) + i;
                xa[i] = jp < dh ? normal_at(jp) : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < H / 2; i++) {
                const int jp = hh * (H / 2) + i;
                xb[i] = jp < dh ? normal_at(dh + jp) : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < H / 2; i++) q += xa[i] * xa[i] + xb[i] * xb[i];
        }
        float ladj = 0.0f, amax = 0.0f;
        for (int c = n_layers - 1; c >= 0; c--) {
            const float* lp = sp + (size_t)c * FD::LAYER;
            if ((c & 1) == 0)
                coupling_layer_hs<H, W, true>(xa, xb, lp, lane, hh, ladj, amax);
            else
                coupling_layer_hs<H, W, true>(xb, xa, lp, lane, hh, ladj, amax);
        }
        q += __shfl_xor(q, 32);
        const float lj = ladj + __shfl_xor(ladj, 32);
        const float am = fmaxf(amax, __shfl_xor(amax, 32));
        const float val = !(am < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
        if (row < n) {
#pragma unroll
            for (int i = 0; i < H / 2; i++) {
                const int jp = hh * (H / 2) + i;
                if (jp < dh) {
                    const float va = xa[i] * scale[jp];
                    const float vb = xb[i] * scale[dh + jp];
                    x[row * d + jp] = (XT)(va + loc[jp]);
                    x[row * d + dh + jp] = (XT)(vb + loc[dh + jp]);
                }
            }
            if (hh == 0) out[row] = (double)val;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// coordinate held in slot i of lane half hh of an autoregressive tile (DL = H / 2 slots per lane): the first DL / 2 slots are
// coordinates hh DL/2 .. of the lower DL coordinates, the other DL / 2 the same range of the upper DL - the order in which the
// coupling tiles hold their two halves, so that the fused pCN step stages both kinds of flow with the same swaps
template <int DL>
__device__ __forceinline__ constexpr int maf_coord(int hh, int i) {
    return (i / (DL / 2)) * DL + hh * (DL / 2) + i % (DL / 2);
}

// Masked autoregressive flow (include/asmc.h, ASMC_FLOW_MAF; reference flows/torch/flows.py:140-168 asks zuko for it by
// default).  A transform is a coupling layer whose conditioner input and transformed block are BOTH the whole x (the MADE masks
// sit in the weights as zeros), so the tile layout and the layer code are the coupling kernels' with H / 2 = the coordinates a
// lane half holds (maf_coord above).  Density: one pass per transform.
template <int H, int W, typename XT, int FLOW_THREADS, bool HS>
__global__ __launch_bounds__(FLOW_THREADS) void k_maf_logprob(int64_t n, int d, const XT* __restrict__ x,
                                                             const float* __restrict__ packed, int n_layers,
                                                             const float* __restrict__ loc, const float* __restrict__ scale,
                                                             float ladj0, float base_const, double* __restrict__ out, int form) {
    extern __shared__ __align__(16) float sp[];
    using FD = FlowDims<H, W>;
    constexpr int FLOW_WAVES = FLOW_THREADS / 64, DL = H / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, hh = lane >> 5;
    if (HS) {
        flow_stage_hs<H, W, FLOW_THREADS>(sp, packed, n_layers);
    } else {
        const int total4 = n_layers * FD::LAYER / 4;
        for (int i4 = threadIdx.x; i4 < total4; i4 += FLOW_THREADS) reinterpret_cast<float4*>(sp)[i4] = reinterpret_cast<const float4*>(packed)[i4];
    }
    __syncthreads();
    const int64_t n_tiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * FLOW_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * FLOW_WAVES) {
        const int64_t row = tile * 32 + p;
        const bool valid = row < n;
        float xv[1][DL];
#pragma unroll
        for (int i = 0; i < DL; i++) {
            const int jp = maf_coord<DL>(hh, i);  // padded coordinates stay at zero: their weights are zero, their (s, t) come out zero
            xv[0][i] = (valid && jp < d) ? flow_standardise((float)x[row * d + jp], loc[jp], scale[jp], 1.0f / scale[jp]) : 0.0f;
        }
        float ladj[1] = {0.0f};
        float amax = 0.0f;
        for (int c = 0; c < n_layers; c++) {
            const float* lp = sp + (size_t)c * FD::LAYER;
            float cond[1][DL];
#pragma unroll
            for (int i = 0; i < DL; i++) cond[0][i] = xv[0][i];
            if constexpr (HS) {  // (the affine form as a compile-time constant of each branch: no per-coordinate branch in the epilogue)
                if (form == 0) coupling_layer_hs<H, W, false, 0>(cond[0], xv[0], lp, lane, hh, ladj[0], amax, 0);
                else coupling_layer_hs<H, W, false, 1>(cond[0], xv[0], lp, lane, hh, ladj[0], amax, 1);
            } else {
                coupling_layer<H, W, 1>(cond, xv, lp, lane, hh, ladj, form);
            }
        }
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < DL; i++) q += xv[0][i] * xv[0][i];
        q += __shfl_xor(q, 32);
        const float lj = ladj[0] + __shfl_xor(ladj[0], 32);
        const float am = fmaxf(amax, __shfl_xor(amax, 32));
        const float val = (HS && !(am < FLOW_HS_MAX)) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
        if (valid && hh == 0) out[row] = (double)val;
    }
}

// Sampling (flows/torch/flows.py:327-346): z ~ N(0, I) from the counter-based generator, the transforms inverted in reverse
// order.  Inverting ONE transform: x_i = z_i exp(s_i(x_<i)) + t_i(x_<i) in the transform's variable order.  The order lives in
// the masks, which the kernel does not see - it iterates x <- z exp(s(x)) + t(x) on ALL coordinates `d` times from x = 0: after
// pass k every coordinate whose inputs have degree < k is final (its inputs were final a pass earlier), so d passes fix them
// all, and the last pass's s is the log-determinant.  Values of not-yet-final coordinates are clamped into the fp16 operand
// range: they reach other coordinates only through masked (zero) weights, and 0 x finite = 0.
// Early exit (round 5): a pass that returns its input bit for bit - every coordinate of every particle of the tile - has found
// the fixed point; every further pass would return the same bits and the same s, so the loop stops there and the result is
// the d-pass result exactly.  A trained flow close to its initialisation (weak couplings) gets there in a few passes; a
// transform with a full-length dependency chain still takes d of them.
// Range: a not-yet-final coordinate (up to +-60000) times a first-layer weight can leave the fp16 range in a HIDDEN
// unit; the inf then meets masked (zero) weights, 0 x inf = NaN lands in coordinates that were already final and the clamp
// would turn it into -60000 without a trace.  So the operand maximum is kept over ALL passes of all transforms and a NaN in
// any pass's output is a failure of its own: either makes the row's log q NaN (rejected and counted like any NaN density).
template <int H, int W, typename XT, int FLOW_THREADS>
__global__ __launch_bounds__(FLOW_THREADS) void k_maf_sample(int64_t n, int d, const float* __restrict__ packed, int n_layers,
                                                            const float* __restrict__ loc, const float* __restrict__ scale,
                                                            float ladj0, float base_const, unsigned long long seed,
                                                            unsigned long long gid0, uint32_t draw_id, XT* __restrict__ x,
                                                            double* __restrict__ out, int all_passes, int form) {
    extern __shared__ __align__(16) float sp[];
    using FD = FlowDims<H, W>;
    constexpr int FLOW_WAVES = FLOW_THREADS / 64, DL = H / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, hh = lane >> 5;
    flow_stage_hs<H, W, FLOW_THREADS>(sp, packed, n_layers);
    __syncthreads();
    const int64_t n_tiles = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * FLOW_WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * FLOW_WAVES) {
        const int64_t row = tile * 32 + p;
        const unsigned long long gid = gid0 + (unsigned long long)(row < n ? row : n - 1);
        float zv[DL], xv[DL];
        float q = 0.0f;
        {   // coordinate c is element c % 4 of quad c / 4 of this particle's draw
            int have = -1;
            double zq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < DL; i++) {
                const int jp = maf_coord<DL>(hh, i);
                if (jp < d && (jp >> 2) != have) {
                    have = jp >> 2;
                    normal_quad_f32(seed, gid, draw_id, (uint32_t)have, zq[0], zq[1], zq[2], zq[3]);
                }
                zv[i] = jp < d ? (float)zq[jp & 3] : 0.0f;
                q += zv[i] * zv[i];
            }
        }
        float ladj = 0.0f, amax = 0.0f;
        for (int c = n_layers - 1; c >= 0; c--) {
            const float* lp = sp + (size_t)c * FD::LAYER;
#pragma unroll
            for (int i = 0; i < DL; i++) xv[i] = 0.0f;
            float ladj_pass = 0.0f;
            for (int pass = 0; pass < d; pass++) {
                float cond[DL], tr[DL];
#pragma unroll
                for (int i = 0; i < DL; i++) cond[i] = xv[i], tr[i] = zv[i];
                ladj_pass = 0.0f;
                float amax_pass = 0.0f;
                coupling_layer_hs<H, W, true>(cond, tr, lp, lane, hh, ladj_pass, amax_pass, form);
                bool same = true, nan = false;
#pragma unroll
                for (int i = 0; i < DL; i++) {
                    same = same && (__float_as_uint(tr[i]) == __float_as_uint(xv[i]));  // (xv is the clamped input: equal bits also mean that no clamp was active)
                    nan = nan || (tr[i] != tr[i]);
                }
                amax = fmaxf(amax, nan ? __builtin_inff() : amax_pass);  // every pass counts, not just the last
#pragma unroll
                for (int i = 0; i < DL; i++) xv[i] = pass + 1 < d ? fminf(fmaxf(tr[i], -60000.0f), 60000.0f) : tr[i];
                if (!all_passes && __all(same)) break;  // (wave uniform) the fixed point: this pass's s was evaluated at final inputs
            }
            ladj += ladj_pass;  // (the final pass: every s is evaluated at final inputs)
#pragma unroll
            for (int i = 0; i < DL; i++) zv[i] = xv[i];
        }
        q += __shfl_xor(q, 32);
        const float lj = ladj + __shfl_xor(ladj, 32);
        const float am = fmaxf(amax, __shfl_xor(amax, 32));
        const float val = !(am < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
        if (row < n) {
#pragma unroll
            for (int i = 0; i < DL; i++) {
                const int jp = maf_coord<DL>(hh, i);
                if (jp < d) x[row * d + jp] = (XT)(zv[i] * scale[jp] + loc[jp]);
            }
            if (hh == 0) out[row] = (double)val;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// flow arithmetic: split-fp16 MFMA (fp32-equivalent operands, asmc_flow_dev.h) unless ASMC_FLOW_MATH=f32 asks for the
// fp32 MFMA chain (read at every launch: a process can compare the two)
bool asmc_flow_math_split() {
    const char* e = getenv("ASMC_FLOW_MATH");
    return !(e && strcmp(e, "f32") == 0);
}

static int flow_half_pad(int dims) { return ((dims / 2 + 15) / 16) * 16; }

static bool flow_supported(int dims, int hidden) {  // (layout 0: 32-particle tiles; dims > 32 take layout 1, asmc_flow16.hip)
    const int H = flow_half_pad(dims);
    return dims <= 32 && dims >= 2 && dims % 2 == 0 && (H == 16 || H == 32) && (hidden == 32 || hidden == 64 || hidden == 128);
}

static int64_t flow_layer_floats(int H, int Wd) { return (int64_t)(2 * (Wd / 32) + H / 16) * 32 + (int64_t)Wd * H + (int64_t)Wd * Wd + 2LL * H * Wd; }

extern "C" int64_t asmc_coupling_pack_floats(int dims, int n_layers, int hidden) {
    if (asmc_flow_layout(ASMC_FLOW_COUPLING, dims, hidden) == 1 && n_layers >= 1)  // 32 < dims <= 128: 16-particle groups (asmc_flow16.hip)
        return asmc_flow16_pack_floats(ASMC_FLOW_COUPLING, dims, n_layers, hidden);
    if (!flow_supported(dims, hidden) || n_layers < 1) {
        asmc_set_error("asmc_coupling_pack_floats: unsupported flow (dims even and <= 64, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    return (int64_t)n_layers * flow_layer_floats(flow_half_pad(dims), hidden);
}

// The packing of `n_layers` conditioner MLPs  dh -> Wd -> Wd -> 2 dh  into MFMA operand order, for lane halves that hold H / 2
// (>= dh / 2 ... padded) inputs each.  A coupling flow of d dims calls it with dh = d / 2 (inputs = the conditioner half,
// outputs = (s_raw, t) of the transformed half); a masked autoregressive flow with dh = d (inputs = outputs = the whole x).
// `maf`: the lane halves' slots hold the coordinates in the autoregressive kernels' order (maf_coord below) instead of hh H/2 + i.
static int flow_pack_layers(const char* who, int H, int dh, int n_layers, int Wd, const float* const* weights_host,
                            const float* const* biases_host, float* packed_host, bool maf = false) {
    // slot i of lane half hh -> coordinate (coupling flows: index inside the conditioner / transformed half)
    auto coord = [&](int hh, int i) { return maf ? (i / (H / 4)) * (H / 2) + hh * (H / 4) + i % (H / 4) : hh * (H / 2) + i; };
    const int NB1 = Wd / 32, NB3 = H / 16;
    if (asmc_flow_math_split()) {  // the split-fp16 layers carry every weight as an fp16 pair
        const int64_t sizes[3] = {(int64_t)Wd * dh, (int64_t)Wd * Wd, (int64_t)2 * dh * Wd};
        for (int c = 0; c < 3 * n_layers; c++)
            for (int64_t k = 0; k < sizes[c % 3]; k++)
                if (!(fabsf(weights_host[c][k]) < 65504.0f)) {
                    asmc_set_error("%s: weight %g of matrix %d is outside the fp16 operand range of the split-fp16 "
                                   "flow kernels (|w| < 65504); ASMC_FLOW_MATH=f32 selects the fp32 MFMA chain", who, (double)weights_host[c][k], c);
                    return ASMC_ERR_UNSUPPORTED;
                }
    }
    const int64_t layer = flow_layer_floats(H, Wd);
    auto unit = [](int s, int hh) { return 32 * (s / 16) + acc_row(s % 16, hh); };  // s-th hidden unit held by half hh
    // packed output row (block nb, row i) -> row of the torch output layer ([s_0..s_dh-1, t_0..t_dh-1]) or -1
    auto out_row = [&](int nb, int i) {
        const int hh = (i / 4) % 2, r = 4 * (i / 8) + i % 4, q = 16 * nb + r;
        const int jp = coord(hh, q < H / 2 ? q : q - H / 2);
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
                    b1[hh * NB1 * 16 + nb * 16 + r] = B1[32 * nb + acc_row(r, hh)];  // [lane half][block][register]: a lane reads its 16 values of a block as four 16-byte reads
                    b2[hh * NB1 * 16 + nb * 16 + r] = B2[32 * nb + acc_row(r, hh)];
                }
        for (int nb = 0; nb < NB3; nb++)
            for (int r = 0; r < 16; r++)
                for (int hh = 0; hh < 2; hh++) {
                    const int o = out_row(nb, acc_row(r, hh));
                    b3[hh * NB3 * 16 + nb * 16 + r] = o >= 0 ? B3[o] : 0.0f;
                }
        for (int nb = 0; nb < NB1; nb++)
            for (int g = 0; g < H / 8; g++)
                for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 4; e++) {
                        const int in = coord(l / 32, 4 * g + e);
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


extern "C" int asmc_coupling_pack(int dims, int n_layers, int hidden, const float* const* weights_host,
                                  const float* const* biases_host, float* packed_host) {
    ASMC_REQUIRE(weights_host && biases_host && packed_host, "null pointer");
    if (asmc_flow_layout(ASMC_FLOW_COUPLING, dims, hidden) == 1 && n_layers >= 1)
        return asmc_flow16_pack(ASMC_FLOW_COUPLING, dims, n_layers, hidden, weights_host, biases_host, packed_host);
    if (!flow_supported(dims, hidden) || n_layers < 1) {
        asmc_set_error("asmc_coupling_pack: unsupported flow (dims even and <= 128, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    return flow_pack_layers("asmc_coupling_pack", flow_half_pad(dims), dims / 2, n_layers, hidden, weights_host, biases_host, packed_host);
}

// ---- masked autoregressive flow: a transform = a coupling layer whose conditioner input and transformed block are both x -----
// H of the layer templates (lane halves hold H / 2 coordinates): 32 for every dims <= 32 - a narrower tile (H = 16 at dims <= 16) would
// halve the first and last layers' work in the stand-alone kernels, but the one-kernel pCN step (asmc_pcn_fused.hip) is built on
// the 32-wide tile, and that step is where the flow is evaluated 32 times per temperature
static int maf_half_pad(int dims) { return dims <= 32 ? 32 : ((dims + 15) / 16) * 16; }
static bool maf_supported(int dims, int hidden) {
    return dims >= 1 && dims <= 32 && (hidden == 32 || hidden == 64 || hidden == 128);
}

extern "C" int64_t asmc_maf_pack_floats(int dims, int n_transforms, int hidden) {
    if (asmc_flow_layout(ASMC_FLOW_MAF, dims, hidden) == 1 && n_transforms >= 1)
        return asmc_flow16_pack_floats(ASMC_FLOW_MAF, dims, n_transforms, hidden);
    if (!maf_supported(dims, hidden) || n_transforms < 1) {
        asmc_set_error("asmc_maf_pack_floats: unsupported flow (dims <= 32, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    return (int64_t)n_transforms * flow_layer_floats(maf_half_pad(dims), hidden);
}

extern "C" int asmc_maf_pack(int dims, int n_transforms, int hidden, const float* const* weights_host,
                             const float* const* biases_host, float* packed_host) {
    ASMC_REQUIRE(weights_host && biases_host && packed_host, "null pointer");
    if (asmc_flow_layout(ASMC_FLOW_MAF, dims, hidden) == 1 && n_transforms >= 1)
        return asmc_flow16_pack(ASMC_FLOW_MAF, dims, n_transforms, hidden, weights_host, biases_host, packed_host);
    if (!maf_supported(dims, hidden) || n_transforms < 1) {
        asmc_set_error("asmc_maf_pack: unsupported flow (dims <= 32, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    return flow_pack_layers("asmc_maf_pack", maf_half_pad(dims), dims, n_transforms, hidden, weights_host, biases_host, packed_host, true);
}

template <int H, int W, typename XT, int FLOW_THREADS, int TPW, bool HS>
static int launch_flow(asmc_ctx* ctx, int64_t n, const XT* x, const asmc_coupling* f, double* out, hipStream_t st) {
    using FD = FlowDims<H, W>;
    const size_t all = (size_t)f->n_layers * FD::LAYER * sizeof(float);
    const int resident = all <= 150 * 1024;
    const size_t lds = resident ? all : (size_t)FD::LAYER * sizeof(float);
    ASMC_REQUIRE(lds <= 160 * 1024, "one coupling layer does not fit in LDS");
    constexpr int FLOW_WAVES = FLOW_THREADS / 64;
    auto kern = k_coupling_logprob<H, W, XT, FLOW_THREADS, TPW, HS>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];  // per instantiation
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t n_tiles = (n + 32 * TPW - 1) / (32 * TPW);
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
    // one tile per wave with next-tile prefetch by default; ASMC_FLOW_TPW=2 selects two tiles per wave (shared A
    // operands, no prefetch) where the accumulators fit the 256-VGPR budget of 2 waves/SIMD
    static const int tpw_env = getenv("ASMC_FLOW_TPW") ? atoi(getenv("ASMC_FLOW_TPW")) : 0;
    const bool hs = asmc_flow_math_split();
#define ASMC_FLOW_CASE(HH, WW)                                                                          \
    if (H == HH && f->hidden == WW) {                                                                   \
        if (HH == 16 && WW <= 64 && tpw_env == 2) return launch_flow<HH, WW, XT, 512, (HH == 16 && WW <= 64) ? 2 : 1, false>(ctx, n, x, f, out, st); \
        if (hs) return launch_flow<HH, WW, XT, 512, 1, true>(ctx, n, x, f, out, st);                    \
        return launch_flow<HH, WW, XT, 512, 1, false>(ctx, n, x, f, out, st);                           \
    }
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

template <int H, int W, typename XT, bool HS>
static int launch_maf(asmc_ctx* ctx, int64_t n, const XT* x, const asmc_coupling* f, double* out, hipStream_t st) {
    using FD = FlowDims<H, W>;
    const size_t lds = (size_t)f->n_layers * FD::LAYER * sizeof(float);
    if (lds > 160 * 1024) {
        asmc_set_error("masked autoregressive flow: %d transforms of width %d do not fit in LDS together (%zu bytes)", (int)f->n_layers, W, lds);
        return ASMC_ERR_UNSUPPORTED;
    }
    auto kern = k_maf_logprob<H, W, XT, 512, HS>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];  // per instantiation
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t want = ((n + 31) / 32 + 7) / 8;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    const int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    ASMC_LAUNCH(ctx, st, "k_maf_logprob", kern, dim3(grid), dim3(512), lds, st, n, (int)f->dims, x, f->packed_dev, (int)f->n_layers,
                f->loc_dev, f->scale_dev, ladj0, base_const, out, (int)f->affine);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

template <typename XT>
static int dispatch_maf(asmc_ctx* ctx, int64_t n, const XT* x, const asmc_coupling* f, double* out, hipStream_t st) {
    if (!maf_supported(f->dims, f->hidden) || f->n_layers < 1) {
        asmc_set_error("masked autoregressive flow: unsupported shape (dims <= 32, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    const int H = maf_half_pad(f->dims);
    const bool hs = asmc_flow_math_split();
#define ASMC_MAF_CASE(HH, WW)                                                              \
    if (H == HH && f->hidden == WW) {                                                      \
        if (hs) return launch_maf<HH, WW, XT, true>(ctx, n, x, f, out, st);                \
        return launch_maf<HH, WW, XT, false>(ctx, n, x, f, out, st);                       \
    }
    ASMC_MAF_CASE(32, 32)
    ASMC_MAF_CASE(32, 64)
    ASMC_MAF_CASE(32, 128)
#undef ASMC_MAF_CASE
    asmc_set_error("masked autoregressive flow: unsupported shape");
    return ASMC_ERR_UNSUPPORTED;
}

template <int H, int W, typename XT>
static int launch_maf_sample(asmc_ctx* ctx, int64_t n, const asmc_coupling* f, unsigned long long seed, unsigned long long gid0,
                             uint32_t draw_id, XT* x, double* out, hipStream_t st) {
    using FD = FlowDims<H, W>;
    const size_t lds = (size_t)f->n_layers * FD::LAYER * sizeof(float);
    if (lds > 160 * 1024) {
        asmc_set_error("asmc_coupling_sample: the flow's transforms must be resident in LDS together");
        return ASMC_ERR_UNSUPPORTED;
    }
    auto kern = k_maf_sample<H, W, XT, 512>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];  // per instantiation
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t want = ((n + 31) / 32 + 7) / 8;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    const int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    ASMC_LAUNCH(ctx, st, "k_maf_sample", kern, dim3(grid), dim3(512), lds, st, n, (int)f->dims, f->packed_dev, (int)f->n_layers,
                f->loc_dev, f->scale_dev, ladj0, base_const, seed, gid0, draw_id, x, out,
                getenv("ASMC_MAF_SAMPLE_ALL_PASSES") ? 1 : 0, (int)f->affine);  // (all_passes: diagnostic, the d-pass loop without the fixed-point exit)
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

extern "C" int asmc_coupling_logprob(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x_dev,
                                     const asmc_coupling* flow, double* out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(flow && x_dev && out_dev, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    ASMC_REQUIRE(flow->packed_dev && flow->loc_dev && flow->scale_dev, "flow parameters missing");
    ASMC_REQUIRE(flow->kind == ASMC_FLOW_COUPLING || flow->kind == ASMC_FLOW_MAF, "bad flow kind");
    ASMC_REQUIRE(flow->affine == ASMC_AFFINE_TANH || (flow->affine == ASMC_AFFINE_SOFTCLIP && flow->kind == ASMC_FLOW_MAF), "bad affine form (the soft-clipped form is for autoregressive flows)");
    hipStream_t st = (hipStream_t)stream;
    if (asmc_flow_layout(flow->kind, flow->dims, flow->hidden) == 1) {
        ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
        ASMC_REQUIRE(flow->n_layers >= 1, "bad n_layers");
        return asmc_flow16_logprob(ctx, n, x_dtype, x_dev, flow, out_dev, st);
    }
    if (flow->kind == ASMC_FLOW_MAF) {
        if (x_dtype == ASMC_F64) return dispatch_maf<double>(ctx, n, (const double*)x_dev, flow, out_dev, st);
        if (x_dtype == ASMC_F32) return dispatch_maf<float>(ctx, n, (const float*)x_dev, flow, out_dev, st);
        asmc_set_error("asmc_coupling_logprob: bad x_dtype");
        return ASMC_ERR_ARG;
    }
    if (!flow_supported(flow->dims, flow->hidden) || flow->n_layers < 1) {
        asmc_set_error("asmc_coupling_logprob: unsupported flow (dims even and <= 64, hidden in {32,64,128})");
        return ASMC_ERR_UNSUPPORTED;
    }
    if (x_dtype == ASMC_F64) return dispatch_flow<double>(ctx, n, (const double*)x_dev, flow, out_dev, st);
    if (x_dtype == ASMC_F32) return dispatch_flow<float>(ctx, n, (const float*)x_dev, flow, out_dev, st);
    asmc_set_error("asmc_coupling_logprob: bad x_dtype");
    return ASMC_ERR_ARG;
}

template <int H, int W, typename XT>
static int launch_flow_sample(asmc_ctx* ctx, int64_t n, const asmc_coupling* f, unsigned long long seed, unsigned long long gid0,
                              uint32_t draw_id, XT* x, double* out, hipStream_t st) {
    using FD = FlowDims<H, W>;
    const size_t lds = (size_t)f->n_layers * FD::LAYER * sizeof(float);
    if (lds > 150 * 1024) {
        asmc_set_error("asmc_coupling_sample: the flow's layers must be resident in LDS together");
        return ASMC_ERR_UNSUPPORTED;
    }
    auto kern = k_coupling_sample<H, W, XT, 512>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];  // per instantiation
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t want = (n_tiles + 7) / 8;
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    const int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    ASMC_LAUNCH(ctx, st, "k_coupling_sample", kern, dim3(grid), dim3(512), lds, st, n, (int)f->dims, f->packed_dev, (int)f->n_layers,
                f->loc_dev, f->scale_dev, ladj0, base_const, seed, gid0, draw_id, x, out);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

extern "C" int asmc_coupling_sample(asmc_ctx* ctx, int64_t n, int x_dtype, const asmc_coupling* flow, uint64_t seed, uint64_t gid0,
                                    uint32_t draw_id, void* x_out_dev, double* lq_out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && flow && x_out_dev && lq_out_dev, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    ASMC_REQUIRE(flow->packed_dev && flow->loc_dev && flow->scale_dev, "flow parameters missing");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(flow->kind == ASMC_FLOW_COUPLING || flow->kind == ASMC_FLOW_MAF, "bad flow kind");
    ASMC_REQUIRE(flow->affine == ASMC_AFFINE_TANH || (flow->affine == ASMC_AFFINE_SOFTCLIP && flow->kind == ASMC_FLOW_MAF), "bad affine form (the soft-clipped form is for autoregressive flows)");
    if (asmc_flow_layout(flow->kind, flow->dims, flow->hidden) == 1)
        return asmc_flow16_sample(ctx, n, x_dtype, flow, seed, gid0, draw_id, x_out_dev, lq_out_dev, (hipStream_t)stream);
    if (flow->kind == ASMC_FLOW_MAF) {
        if (!maf_supported(flow->dims, flow->hidden) || flow->n_layers < 1 || !asmc_flow_math_split()) {
            asmc_set_error("asmc_coupling_sample: unsupported autoregressive flow shape, or the fp32 MFMA chain was asked for (split-fp16 layers only)");
            return ASMC_ERR_UNSUPPORTED;
        }
        hipStream_t stm = (hipStream_t)stream;
        const int Hm = maf_half_pad(flow->dims);
#define ASMC_MAF_SAMPLE_CASE(HH, WW)                                                                                      \
    if (Hm == HH && flow->hidden == WW) {                                                                                 \
        if (x_dtype == ASMC_F64)                                                                                          \
            return launch_maf_sample<HH, WW, double>(ctx, n, flow, seed, gid0, draw_id, (double*)x_out_dev, lq_out_dev, stm); \
        return launch_maf_sample<HH, WW, float>(ctx, n, flow, seed, gid0, draw_id, (float*)x_out_dev, lq_out_dev, stm);   \
    }
        ASMC_MAF_SAMPLE_CASE(32, 32)
        ASMC_MAF_SAMPLE_CASE(32, 64)
        ASMC_MAF_SAMPLE_CASE(32, 128)
#undef ASMC_MAF_SAMPLE_CASE
        asmc_set_error("asmc_coupling_sample: unsupported autoregressive flow shape");
        return ASMC_ERR_UNSUPPORTED;
    }
    if (!flow_supported(flow->dims, flow->hidden) || flow->n_layers < 1 || !asmc_flow_math_split()) {
        asmc_set_error("asmc_coupling_sample: unsupported flow shape, or the fp32 MFMA chain was asked for (split-fp16 layers only)");
        return ASMC_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int H = flow_half_pad(flow->dims);
#define ASMC_SAMPLE_CASE(HH, WW)                                                                                          \
    if (H == HH && flow->hidden == WW) {                                                                                  \
        if (x_dtype == ASMC_F64)                                                                                          \
            return launch_flow_sample<HH, WW, double>(ctx, n, flow, seed, gid0, draw_id, (double*)x_out_dev, lq_out_dev, st); \
        return launch_flow_sample<HH, WW, float>(ctx, n, flow, seed, gid0, draw_id, (float*)x_out_dev, lq_out_dev, st);   \
    }
    ASMC_SAMPLE_CASE(16, 32)
    ASMC_SAMPLE_CASE(16, 64)
    ASMC_SAMPLE_CASE(16, 128)
    ASMC_SAMPLE_CASE(32, 32)
    ASMC_SAMPLE_CASE(32, 64)
    ASMC_SAMPLE_CASE(32, 128)
#undef ASMC_SAMPLE_CASE
    asmc_set_error("asmc_coupling_sample: unsupported flow shape");
    return ASMC_ERR_UNSUPPORTED;
}
