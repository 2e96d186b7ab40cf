// asmc_pcn_fused.hip — the flow-proposal pCN step as ONE kernel.
#include <stdlib.h>

#include "asmc_common.h"
#include <type_traits>
#include "asmc_pcn_dev.h"
#include "asmc_flow_dev.h"

#ifndef FUSED_THREADS
#define FUSED_THREADS 512  // 8 waves per CU: two per SIMD (the weights in LDS allow one block per CU)
#endif

// =============================================================================================================
// Fused flow-proposal pCN step (reference smc/minipcn.py:97-114 around smc/base.py:507-519 with
// flows/torch/flows.py:368-387 as log q): propose -> coupling flow on the fp32 MFMA -> built-in targets -> accept in ONE
// kernel.  x' never touches HBM, y' is not regenerated, the step is one launch instead of three.
//
// A wave takes 64 particles at a time from a device-side tile counter (no ragged last round).  Phase 1 is the
// register-resident proposal of k_pcn_reg_flow: one lane per particle, y read coordinate-major, y' = a y + rho xi,
// x' = mu + L y' four rows at a time with wave-uniform coefficients (scalar loads), folded into the two quadratic forms
// of the targets and converted to the flow's standardised fp32 input on the fly.  The flow kernel wants 32 particles
// per tile with the two lane halves splitting each particle's coordinates: ONE v_permlane32_swap per register pair
// (coordinate j of the lower coordinate block, coordinate j + 8 of the upper one) turns the 64 lane-private rows into
// the operands of TWO flow tiles at once - its first result is tile A (particles 0-31: own value in the lower half, the
// partner block from lane l - 32 in the upper half), its second tile B (particles 32-63).  After the four coupling
// layers both halves of a tile hold log q, so lane l < 32 reads tile A's and lane l >= 32 tile B's: no shuffle back.
// Phase 3 is the accept step on the lane's own particle: y <- y' for accepted lanes only.  y' does not wait in registers
// for that (64 VGPRs next to the flow's accumulators would not fit two waves per SIMD): phase 1 parks it in a scratch
// buffer of the state's layout (fire-and-forget stores), accepted lanes fetch it back (cache-hot) and store it into the state.
// The flow's weights stay resident in LDS (115 KB at d = 32, W = 64: one block of 8 waves per CU, two waves per SIMD, so
// one wave's vector work - noise, mat-vec, accept - runs in the shadow of its partner's MFMA chains).
#define FUSED_TL_DOUBLES 1368  // doubles of the pCN tables in LDS (d = 32), 16-byte aligned end: the larger of the two layouts below
#define FUSED_MAX_COMPONENTS 4  // mixture components per built-in target in the matrix-core variant
#ifndef MV_DEPTH
#define MV_DEPTH 4  // batches of mat-vec coefficients in flight (16 VGPRs each)
#endif
typedef double double2v __attribute__((ext_vector_type(2)));
// row group of batch bi: the groups' batches are laid end to end, group g has 2 g + 2 of them (g (g + 1) in front of it)
__host__ __device__ constexpr int mv_group(int bi) {
    int g = 0;
    while ((g + 1) * (g + 2) <= bi) g++;
    return g;
}
template <int I, int N, typename F>
__device__ __forceinline__ void mv_for_each(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        mv_for_each<I + 1, N>(f);
    }
}
// FUSED_INPLACE (round 4): the proposal y' stays in the register file across the flow and ACCEPTED lanes store it into the state
// in place - one state array, no park, no copy of the minority, HBM traffic = the algorithmic 2 d s + 16 bytes per particle at
// any acceptance rate.  The flow then runs one 32-particle tile at a time (coupling_layer_hs1p: 64 accumulator registers instead
// of the 128 of the two interleaved tiles, which is where y' lives).  FUSED_INPLACE=0: round 3's two-halves state with a parity
// byte per tile and the two tiles interleaved (coupling_layer_hs2).
#ifndef FUSED_INPLACE
#define FUSED_INPLACE 1
#endif
// FUSED_MVMFMA (round 4): x' = mu + L y' on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) instead of 528 vector FMAs per
// particle fed by 288 LDS broadcast reads per tile (the mat-vec was LDS-latency bound: 11 k of a lone wave's 61 k cycles per tile
// for 2 k cycles of arithmetic).  The 64 lane-private rows of y' are turned into MFMA operands IN PLACE by a 4 x 4 transpose of
// (register within a K-step, 16-lane group) - two swap stages, v_permlane32_swap then v_permlane16_swap - after which register
// 4 s + r holds, in lane (g, n), coordinate 4 s + g of particle 16 r + n: the B operand of K-step s for particle block r.  The
// coefficient image (A operand: one double per lane per (row block, K-step), zero above the diagonal) costs 12 LDS reads per
// tile.  A lane receives rows 16 mb + 4 g + r of its four particle blocks; the quadratic forms are summed over a particle's four
// lanes by the same transpose (+ 3 adds), the flow's standardised input is formed in place and ONE v_permlane16_swap per
// register pair lands it in the flow tiles' layout (lane half hh: coordinates 8 hh .. 8 hh + 7 of either half - the layout the
// permlane32 swaps produced).  The accepted lanes' y' is stored from the transposed registers (each still a run of 16
// consecutive particles of one coordinate: 128-byte segments).  Coupling flows only; FUSED_INPLACE only.
#ifndef FUSED_MVMFMA
#define FUSED_MVMFMA FUSED_INPLACE
#endif
typedef double fused_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fused_swap32(double& x, double& y) {  // x.lanes[32..63] <-> y.lanes[0..31]
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x), yb = __builtin_bit_cast(unsigned long long, y);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)xb, (unsigned)yb, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
    y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ void fused_swap16(double& x, double& y) {  // the odd 16-lane rows of x <-> the even rows of y
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x), yb = __builtin_bit_cast(unsigned long long, y);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
    y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
// (q0, q1, q2, q3)[lane (g, n)] -> (q_g of lane (0, n), q_g of lane (1, n), q_g of lane (2, n), q_g of lane (3, n)): the 4 x 4
// transpose of (register, 16-lane group)
__device__ __forceinline__ void fused_transpose4(double& q0, double& q1, double& q2, double& q3) {
    fused_swap32(q0, q2);
    fused_swap32(q1, q3);
    fused_swap16(q0, q1);
    fused_swap16(q2, q3);
}
#ifndef HS1P_PREFETCH
#define HS1P_PREFETCH false  // (true: the A operands of the next K-step group are read while the current group's MFMAs issue - 16 more registers, measured: no gain)
#endif
#ifndef FUSED_PRIO_A
#define FUSED_PRIO_A 2
#define FUSED_PRIO_B 1
#endif
#ifndef MV_CHUNK
#define MV_CHUNK 8  // columns of L between two ordering points of the mat-vec
#endif
// KIND: ASMC_FLOW_COUPLING - the coupling layers described above; ASMC_FLOW_MAF - masked autoregressive transforms
// (flows/torch/flows.py:140-168, the reference's default flow class): a transform is a coupling layer whose conditioner input and
// transformed block are both the whole x (asmc_flow.hip), so a flow tile holds 16 coordinates per lane instead of 8 + 8 and the
// swap pairs coordinate i with coordinate 16 + i.
// MIX: built-in MIXTURE targets (2 .. FUSED_MAX_COMPONENTS components) - their own instantiations: the component loop keeps the
// mat-vec accumulators alive past the first quadratic forms, and compiled into the single-Gaussian kernel it cost the headline
// 90 spilled registers (0.294 -> 0.320 ms per step).
template <typename T, int W, int NOISE, bool HS, int KIND = ASMC_FLOW_COUPLING, bool MIX = false>
__global__ __launch_bounds__(FUSED_THREADS) void k_pcn_flow_fused(
    int64_t n, double* __restrict__ ll, double* __restrict__ lp, double* __restrict__ lq, const double* __restrict__ ptab,
    PcnScalars p, const double* rho_ptr, uint32_t step, const float* __restrict__ packed, int n_layers,
    const float* __restrict__ loc, const float* __restrict__ scale, float ladj0, float base_const,
    unsigned int* __restrict__ tile_counter, long long* __restrict__ block_counts, PcnAdaptArgs ad, int par_words, int affine) {
    constexpr int D = 32, H = 16, THREADS = FUSED_THREADS;
    constexpr int HF = KIND == ASMC_FLOW_MAF ? 32 : H;  // H of the flow's layer templates (lane halves hold HF / 2 inputs)
    static_assert(KIND == ASMC_FLOW_COUPLING || HS, "the autoregressive variant runs the split-fp16 layers only");
    using FD = FlowDims<HF, W>;
    extern __shared__ __align__(16) float sp[];
    if (HS) {
        flow_stage_hs<HF, W, THREADS>(sp, packed, n_layers);  // split-fp16 operand images (asmc_flow_dev.h)
    } else {  // flow weights -> LDS; all of a thread's loads are issued before its first LDS store
        const int total4 = n_layers * FD::LAYER / 4;
        for (int base = 0; base < total4; base += THREADS * 8) {
            float4 tmp[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i4 = base + q * THREADS + threadIdx.x;
                tmp[q] = i4 < total4 ? reinterpret_cast<const float4*>(packed)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i4 = base + q * THREADS + threadIdx.x;
                if (i4 < total4) reinterpret_cast<float4*>(sp)[i4] = tmp[q];
            }
        }
    }
    // pCN tables behind the weights, read by every lane at the same address (LDS broadcast): the dense lower triangle of
    // L (row stride D, so that two consecutive coefficients are one aligned 16-byte read), mu, the two targets' mean /
    // precision rows and constants, the flow's loc / scale.  (Scalar loads would be the natural home of wave-uniform
    // coefficients, but inside this kernel's tile loop LLVM issues all ~1400 of them up front and spills the SGPRs.)
    double* tl = reinterpret_cast<double*>(sp + (size_t)n_layers * FD::LAYER);
    constexpr int T_MU = D * D, T_LLMU = T_MU + D, T_LLPR = T_LLMU + D, T_LPMU = T_LLPR + D, T_LPPR = T_LPMU + D,
                  T_LOGW = T_LPPR + D, T_LOC = T_LOGW + 2;  // then loc / scale / 1/scale: 3 x D floats = 1.5 D doubles
    static_assert(T_LOC + 3 * D / 2 <= FUSED_TL_DOUBLES, "pCN tables");
    // matrix-core variant: operand image of L | mu | per target t (0 = likelihood, 1 = prior) and component c: mean, precision |
    // log-weights [t * 4 + c] | loc / scale / 1/scale (floats); per-row tables in the lanes' reading order (below)
    constexpr int M_MU = 12 * 64, M_MIX = M_MU + D, M_LOGW = M_MIX + 2 * FUSED_MAX_COMPONENTS * 2 * D, M_LOC = M_LOGW + 2 * FUSED_MAX_COMPONENTS;
    static_assert(M_LOC + 3 * D / 2 <= FUSED_TL_DOUBLES, "pCN tables (matrix-core variant)");
    constexpr bool MVM = FUSED_MVMFMA && FUSED_INPLACE;  // the mat-vec on the fp64 matrix cores
    {
        const double* m0g = ptab + 2 * PTAB_TRI(D) + D;
        // A problem of fewer than 32 dimensions arrives zero-padded to 32 (asmc_pcn_mutate_flow: tables with the identity beyond
        // dr = p.d_noise, state rows padded, no noise on the padding).  y keeps its natural order; the ROWS of x' are placed where
        // the flow's tiles want them: a coupling flow of dr dims splits x at dr / 2 and holds each half in 16 slots, so row r < 16
        // is coordinate r of the first half (r < dr / 2) and row 16 + r coordinate dr / 2 + r of the second - both still below the
        // diagonal of L's natural order, so the triangular K-step pattern holds; an autoregressive flow keeps the natural order.
        // Padding rows have no coefficients, no mean, no weight in the targets and loc = 0, scale = 1.
        const int dr = (p.d_noise > 0 && p.d_noise < D) ? p.d_noise : D;
        const int dhalf = KIND == ASMC_FLOW_MAF ? 16 : dr / 2;
        auto nat = [&](int r) -> int {  // row of the 32-row layout -> natural coordinate, or -1 for padding
            const int c = r < 16 ? (r < dhalf ? r : -1) : (r - 16 + dhalf);
            return (c >= 0 && c < dr) ? c : -1;
        };
        if (MVM) {
            // A-operand image of L: slot (mb, s) = mb == 0 ? s : 4 + s (row block mb, K-step s; s < 4 mb + 4), 64 doubles each:
            // lane l supplies A[i = l & 15][k = l >> 4]; accumulator row i = h + 4 r lands in lane group h, register r, and is made
            // to BE row 16 mb + 4 h + r of x' by storing that row of L at image row i = (row % 4) * 4 ... i.e. i % 4 = h, i / 4 = r
            for (int e = threadIdx.x; e < 12 * 64; e += THREADS) {
                const int slot = e >> 6, l = e & 63, mb = slot < 4 ? 0 : 1, sidx = slot < 4 ? slot : slot - 4;
                const int i = l & 15, j = nat(16 * mb + 4 * (i & 3) + (i >> 2)), k = 4 * sidx + (l >> 4);
                tl[e] = (j >= 0 && k <= j) ? ptab[j * (j + 1) / 2 + k] : 0.0;
            }
            // per-row tables in the order a lane reads them: entry h * 8 + mb * 4 + r <-> row 16 mb + 4 h + r
            for (int e = threadIdx.x; e < D; e += THREADS) {
                const int j = nat(16 * ((e >> 2) & 1) + 4 * (e >> 3) + (e & 3));
                tl[M_MU + e] = j >= 0 ? ptab[2 * PTAB_TRI(D) + j] : 0.0;
                reinterpret_cast<float*>(tl + M_LOC)[e] = j >= 0 ? loc[j] : 0.0f;
                reinterpret_cast<float*>(tl + M_LOC)[D + e] = j >= 0 ? scale[j] : 1.0f;
                reinterpret_cast<float*>(tl + M_LOC)[2 * D + e] = j >= 0 ? 1.0f / scale[j] : 1.0f;
            }
            for (int e = threadIdx.x; e < 2 * FUSED_MAX_COMPONENTS * D; e += THREADS) {
                const int tc = e / D, er = e % D, tt = tc / FUSED_MAX_COMPONENTS, c = tc % FUSED_MAX_COMPONENTS;
                const int j = nat(16 * ((er >> 2) & 1) + 4 * (er >> 3) + (er & 3));
                const bool live = j >= 0 && c < (tt == 0 ? p.c_ll : p.c_lp);
                const double* mg = m0g + (size_t)tt * PTAB_MIX(D);
                tl[M_MIX + tc * 2 * D + er] = live ? mg[ASMC_MAX_COMPONENTS + c * D + j] : 0.0;
                tl[M_MIX + tc * 2 * D + D + er] = live ? mg[ASMC_MAX_COMPONENTS * (1 + D) + c * D + j] : 0.0;
            }
            if (threadIdx.x < 2 * FUSED_MAX_COMPONENTS) {
                const int tt = threadIdx.x / FUSED_MAX_COMPONENTS, c = threadIdx.x % FUSED_MAX_COMPONENTS;
                tl[M_LOGW + threadIdx.x] = c < (tt == 0 ? p.c_ll : p.c_lp) ? m0g[(size_t)tt * PTAB_MIX(D) + c] : -INFINITY;
            }
        } else {
            for (int e = threadIdx.x; e < D * D; e += THREADS) {
                const int j = e / D, k = e - j * D;
                tl[e] = k <= j ? ptab[j * (j + 1) / 2 + k] : 0.0;
            }
            for (int e = threadIdx.x; e < D; e += THREADS) {
                tl[T_MU + e] = ptab[2 * PTAB_TRI(D) + e];
                tl[T_LLMU + e] = m0g[ASMC_MAX_COMPONENTS + e];
                tl[T_LLPR + e] = m0g[ASMC_MAX_COMPONENTS * (1 + D) + e];
                tl[T_LPMU + e] = m0g[PTAB_MIX(D) + ASMC_MAX_COMPONENTS + e];
                tl[T_LPPR + e] = m0g[PTAB_MIX(D) + ASMC_MAX_COMPONENTS * (1 + D) + e];
                reinterpret_cast<float*>(tl + T_LOC)[e] = loc[e];
                reinterpret_cast<float*>(tl + T_LOC)[D + e] = scale[e];
                reinterpret_cast<float*>(tl + T_LOC)[2 * D + e] = 1.0f / scale[e];
            }
        }
        if (!MVM && threadIdx.x == 0) tl[T_LOGW] = m0g[0], tl[T_LOGW + 1] = m0g[PTAB_MIX(D)];
    }
    // ... and the Box-Muller tables of the default noise behind them (6 KB)
    bm_d2* bmt = reinterpret_cast<bm_d2*>(tl + FUSED_TL_DOUBLES);
    if (NOISE == ASMC_NOISE_F64) bm_tab_stage<THREADS>(bmt, p.bmtab);
    // ... and the tiles' parity bits (below), one bit per tile, when they fit (par_words > 0): a snapshot is all a launch
    // needs - a tile is read once per launch, before its own wave may flip it
    unsigned* const par_bits = reinterpret_cast<unsigned*>(bmt + BM_TAB_N);
    for (int wd = threadIdx.x; wd < par_words; wd += THREADS) {
        const uint4 b0 = reinterpret_cast<const uint4*>(p.tile_par)[2 * wd], b1 = reinterpret_cast<const uint4*>(p.tile_par)[2 * wd + 1];
        const unsigned q[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {  // bytes (0 / 1) of a dword -> 4 bits: the product lines bit 0 of byte m up at bit 24 + m
            const unsigned nib = (((q[k] & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
            bits |= nib << (4 * k);
        }
        par_bits[wd] = bits;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5;
    const bool first_on_simd = __builtin_amdgcn_readfirstlane(wave) < 4;
#ifdef FUSED_STAGGER  // experiment (MI355X_MICROARCH.md, two waves per SIMD, item 9): the second wave of every SIMD starts late
    if (!first_on_simd) {
#pragma unroll 1
        for (int i = 0; i < FUSED_STAGGER; i++) __builtin_amdgcn_s_sleep(127);  // 127 x 64 cycles each
    }
#endif
    double rho;
    if (ad.prev_cell) {  // sharded, t > 0: close step t-1 here (see PcnAdaptArgs)
        const double rp = ad.rho_hist[ad.t - 1];
        if (ad.adapt <= 1) {
            const long long cp = *ad.prev_cell;
            rho = ad.adapt ? pcn_adapt_rho(rp, cp, ad.n, ad.target, ad.t - 1) : rp;
            if (blockIdx.x == 0 && threadIdx.x == 0) ad.counts_out[ad.t - 1] = cp;
        } else if (ad.t % ad.adapt == 0) {  // lagged: the block [t - k, t - 1] was exchanged as a whole (cells [t' % k])
            double r = rp;
            for (int tp = ad.t - ad.adapt; tp < ad.t; tp++) {
                const long long ct = ad.prev_cell[tp % ad.adapt];
                r = pcn_adapt_rho(r, ct, ad.n, ad.target, tp);
                if (blockIdx.x == 0 && threadIdx.x == 0) ad.counts_out[tp] = ct;
            }
            rho = r;
        } else {
            rho = rp;  // inside a block: the held step size
        }
    } else {
        rho = *rho_ptr;
    }
    if (ad.cell && blockIdx.x == 0 && threadIdx.x == 0) ad.rho_hist[ad.t] = rho;
    const double a = sqrt(1.0 - rho * rho);
    const int dn = __builtin_amdgcn_readfirstlane((p.d_noise > 0 && p.d_noise < 32) ? p.d_noise : 32);  // real dimension of a zero-padded problem
    const int64_t n_tiles = (n + 63) / 64;
    long long n_acc = 0, n_bad = 0;
#ifdef FUSED_STAMP
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    int ntile = 0;
#define STAMP(k)                                                  \
    do {                                                          \
        __builtin_amdgcn_sched_barrier(0);                        \
        unsigned long long tn_;                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory"); \
        tsum[k] += tn_ - tprev;                                   \
        tprev = tn_;                                              \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)
#else
#define STAMP(k)
#endif
    // coordinate-major state through one buffer descriptor (see soa_load / soa_store)
    const unsigned long long ysa = (unsigned long long)(uintptr_t)p.ys;
    T* ysu = reinterpret_cast<T*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ysa >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ysa));
    // The allocation holds TWO copies of the state's layout.  Where a tile's current state lives is the tile's parity byte
    // (p.tile_par, zeroed when the state is whitened into half 0): a step reads y from its half A = half[par], parks y' in
    // the other half B before the flow, and after the accept test either copies the REJECTED lanes' y from A to B and
    // flips the parity, or the ACCEPTED lanes' y' from B to A and leaves it - whichever moves fewer lanes (none at all
    // when every lane accepted).  One store of the state per step instead of park + re-read + store: HBM traffic per launch
    // 1.04 -> 0.55 GB at 98 % acceptance (round 2 parked in half 1 and always copied the accepted lanes back).
    const int half_records = (int)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)p.n_pad * D * sizeof(T)));
    T* const ysu1 = ysu + (size_t)__builtin_amdgcn_readfirstlane((int)(unsigned)p.n_pad) * D;
    const unsigned ys_row = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p.n_pad) * (unsigned)sizeof(T);
    const unsigned ys_lane = (unsigned)lane * (unsigned)sizeof(T);
    unsigned char* __restrict__ const tile_par = p.tile_par;
#ifdef FUSED_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
#ifdef FUSED_HALF
    const bool idle_wave = wave >= 4;  // diagnostic: one wave per SIMD, so the stamps show a wave's own time per phase
#else
    const bool idle_wave = false;
#endif
    // Software pipeline over the tiles a wave takes from the device-side counter: the NEXT tile's index is requested when
    // the flow phase starts (no memory instruction in that phase waits behind it), and its state loads are issued right
    // after the flow, in front of this tile's accept step - so a tile's counter round trip, parity lookup and 32 state
    // loads no longer sit one after the other in front of its first instruction.
#ifdef FUSED_STATIC_TILES  // diagnostic: tiles dealt round-robin, no counter (what the hand-out itself costs)
    unsigned static_next = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    auto tile_fetch = [&]() -> unsigned {
        const unsigned t_l = static_next;
        static_next += gridDim.x * (THREADS / 64);
        return t_l;
    };
#else
    // Tiles are DEALT round-robin for as many whole rounds as there are (wave w of the grid takes tiles w, w + W, w + 2 W, ...);
    // only the remainder - fewer tiles than waves - goes through the device-side counter, to whichever waves get there first.
    // Round 3 drew every tile from the counter: 15 625 returning atomics on one word per launch run the memory-side atomic unit
    // at 60 % of what it can serve (MI355X_MICROARCH.md, dequeue), and the waits behind it cost the step 4 % (0.314 -> 0.302 ms
    // with no counter at all; grouped requests had cost 3-4 % in tail imbalance).
    const unsigned deal_stride = gridDim.x * (unsigned)(THREADS / 64);
    const unsigned n_dealt = (unsigned)(((n + 63) / 64) / deal_stride) * deal_stride;
    unsigned deal_next = blockIdx.x * (unsigned)(THREADS / 64) + (threadIdx.x >> 6);
    auto tile_fetch = [&]() -> unsigned {
        if (deal_next < n_dealt) {  // (wave uniform)
            const unsigned t_d = deal_next;
            deal_next += deal_stride;
            return t_d;
        }
        unsigned t_l = 0;
        if (lane == 0) t_l = n_dealt + atomicAdd(tile_counter, 1u);
        return t_l;
    };
#endif
    auto tile_parity = [&](unsigned t) -> unsigned {
        if (FUSED_INPLACE) return 0u;  // one state array
        if (par_words > 0) return (unsigned)__builtin_amdgcn_readfirstlane((int)((par_bits[t >> 5] >> (t & 31u)) & 1u));
        return (unsigned)__builtin_amdgcn_readfirstlane((int)tile_par[t]);
    };
    auto tile_load = [&](unsigned t, unsigned par, double(&v)[D], double& oll, double& olp, double& olq) __attribute__((always_inline)) {
        const int64_t i = (int64_t)t * 64 + lane;
        const bool valid = i < n;
        const unsigned ys_tile = t * 64u * (unsigned)sizeof(T);
        const __amdgpu_buffer_rsrc_t ysr = __builtin_amdgcn_make_buffer_rsrc(par ? ysu1 : ysu, 0, half_records, 0x00020000);  // A
        // (unconditional: a tile's 64 slots exist - the state is padded to whole tiles and zeroed when allocated; a ragged last
        // tile's spare lanes carry finite leftovers that no other lane ever sees and `valid` keeps out of every result)
        (void)valid;
#pragma unroll
        for (int j = 0; j < D; j++) v[j] = soa_load<T>(ysr, ys_lane, ys_tile + (unsigned)j * ys_row);
        oll = olp = olq = 0.0;
        if (valid) oll = ll[i], olp = lp[i], olq = lq[i];
    };
    double v[D];
    double oll = 0.0, olp = 0.0, olq = 0.0;
    unsigned t = idle_wave ? 0xFFFFFFFFu : (unsigned)__builtin_amdgcn_readfirstlane((int)tile_fetch());
    unsigned par = 0;
    bool have = (int64_t)t < n_tiles;
    if (have) {
        par = tile_parity(t);
        tile_load(t, par, v, oll, olp, olq);
    }
    while (have) {
        STAMP(0);
        // the tables are loop invariant: an offset LLVM cannot see through keeps their reads inside the tile loop (hoisted,
        // they would need a thousand registers)
        // (a VECTOR register: one base address + 16-bit immediate offsets reach every table entry; with a scalar offset the
        // addresses are formed on the scalar side and each far read pays a v_mov - 350 of them per tile)
        int zoff;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
        const double* __restrict__ Lt = tl + zoff;
        const float* __restrict__ locs = reinterpret_cast<const float*>(tl + (MVM ? M_LOC : T_LOC)) + zoff;
        const int64_t i = (int64_t)t * 64 + lane;
        const bool valid = i < n;
        const unsigned ys_tile = t * 64u * (unsigned)sizeof(T);
#if !FUSED_INPLACE
        const __amdgpu_buffer_rsrc_t ypr = __builtin_amdgcn_make_buffer_rsrc(par ? ysu : ysu1, 0, half_records, 0x00020000);  // B
#endif
        const unsigned long long gid = p.gid0 + (unsigned long long)i;
        // ---- phase 1: proposal, one lane per particle (y, ll, lp, lq: loaded behind the previous tile's flow) ---------
        // |y|^2 is accumulated coordinate by coordinate as the noise loop consumes y (same chain, same bits): summed up front it
        // made the tile's first instruction wait for ALL 32 state loads - they now land behind the first noise quads.  Only the
        // Student-t reference needs the sum before the first proposal coordinate (its scale depends on it).
        double q0 = 0.0, q1 = 0.0, q0t = 0.0;
        if (p.gam != nullptr) {  // (tpcn_scale's own test)
#pragma unroll
            for (int j = 0; j < D; j++) q0t = fma(v[j], v[j], q0t);
        }
        STAMP(1);
        const double rs = tpcn_scale(rho, p.nu, q0t, p.gam, valid ? i : 0);
        // The real dimension of a zero-padded problem (dn < D) is a RUN-TIME uniform: tested per quad it is two branches per quad and, where
        // the paths meet, copies of the running sums and of the quad's coordinates (six 64-bit moves per quad in the generated code of
        // round 5) - and every branch ends a scheduling region.  One branch per tile instead: the full-width loop carries no test at all
        // (round 6; the same finding as the flows' affine form).
        auto noise_phase = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int qd = 0; qd < D / 4; qd++) {
                if (!FULL && 4 * qd >= dn) continue;  // (wave uniform) a zero-padded problem: no noise beyond its dimension - y stays 0 there
                double z[4];
                if constexpr (NOISE == ASMC_NOISE_F64) normal_quad(p.seed, gid, step, (uint32_t)qd, bmt, z[0], z[1], z[2], z[3]);
                else normal_quad_f32(p.seed, gid, step, (uint32_t)qd, z[0], z[1], z[2], z[3]);
                if (FULL || 4 * qd + 4 <= dn) {  // (wave uniform; the common case: no per-element test, no selects)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        q0 = fma(v[4 * qd + e], v[4 * qd + e], q0);
                        v[4 * qd + e] = (double)(T)fma(rs, z[e], a * v[4 * qd + e]);
                        q1 = fma(v[4 * qd + e], v[4 * qd + e], q1);
                    }
                } else {  // the quad that straddles the dimension of a zero-padded problem
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (4 * qd + e >= dn) continue;
                        q0 = fma(v[4 * qd + e], v[4 * qd + e], q0);
                        v[4 * qd + e] = (double)(T)fma(rs, z[e], a * v[4 * qd + e]);
                        q1 = fma(v[4 * qd + e], v[4 * qd + e], q1);
                    }
                }
#ifndef FUSED_NOISE_SB
#define FUSED_NOISE_SB 1  // quads between two scheduling barriers of the noise phase
#endif
                if constexpr (NOISE == ASMC_NOISE_F64) {
                    if (qd % FUSED_NOISE_SB == FUSED_NOISE_SB - 1) __builtin_amdgcn_sched_barrier(0);
                } else {
                    if (qd & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (dn == D) noise_phase(std::true_type{});
        else noise_phase(std::false_type{});
        STAMP(2);
        // x'_j = mu_j + sum_k L[j,k] y'_k, four rows at a time: straight into the targets' quadratic forms and into the
        // flow's standardised fp32 input (the same (float) x' and division the stand-alone flow kernel applies)
        float xf[D];
        double qa = 0.0, qb = 0.0, mix_ll = 0.0, mix_lp = 0.0;
        if constexpr (MVM) {
            // y' -> MFMA operands, in place: register 4 s + r <- (coordinate 4 s + g, particle 16 r + n) in lane (g, n)
#pragma unroll
            for (int sI = 0; sI < D / 4; sI++) fused_transpose4(v[4 * sI], v[4 * sI + 1], v[4 * sI + 2], v[4 * sI + 3]);
            const double* __restrict__ rowt = Lt + (lane >> 4) * 8;  // this lane group's rows: entry mb * 4 + r <-> row 16 mb + 4 g + r
            const double* __restrict__ Aimg = Lt + lane;
            fused_d4 acc[2][4];
            // (the per-row tables are read as 16-byte vectors: `ds_read_b128` takes a 16-bit byte offset, the `ds_read2_b64` the compiler
            // picks for scalar doubles only 8 bits of 8 bytes - every read beyond 2 KB of the lane's base cost one `v_add_u32`, 33 per tile)
            typedef double row_d2 __attribute__((ext_vector_type(2)));
            typedef float row_f2 __attribute__((ext_vector_type(2)));
            static_assert(M_MU % 2 == 0 && M_MIX % 2 == 0 && D % 4 == 0, "16-byte aligned row tables");
            const row_d2* __restrict__ row2 = reinterpret_cast<const row_d2*>(rowt);
            auto row4 = [&](int off, int mb, double (&out)[4]) {
                const row_d2 lo = row2[(off + 4 * mb) / 2], hi = row2[(off + 4 * mb) / 2 + 1];
                out[0] = lo.x, out[1] = lo.y, out[2] = hi.x, out[3] = hi.y;
            };
#pragma unroll
            for (int mb = 0; mb < 2; mb++) {
                double mu4[4];
                row4(M_MU, mb, mu4);
                const fused_d4 m = {mu4[0], mu4[1], mu4[2], mu4[3]};
#pragma unroll
                for (int nb = 0; nb < 4; nb++) acc[mb][nb] = m;
            }
#pragma unroll
            for (int sI = 0; sI < D / 4; sI++) {
                const double a1 = Aimg[(4 + sI) * 64];
                const double a0 = sI < 4 ? Aimg[sI * 64] : 0.0;
#pragma unroll
                for (int nb = 0; nb < 4; nb++) {
                    if (sI < 4) acc[0][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, v[4 * sI + nb], acc[0][nb], 0, 0, 0);
                    acc[1][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, v[4 * sI + nb], acc[1][nb], 0, 0, 0);
                }
                if (sI & 1) __builtin_amdgcn_sched_barrier(0);  // (operand reads stay next to their K-steps)
            }
            double qpa[4] = {0.0, 0.0, 0.0, 0.0}, qpb[4] = {0.0, 0.0, 0.0, 0.0};
            const float* __restrict__ locr = locs + (lane >> 4) * 8;
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int rp = 0; rp < 2; rp++) {  // two rows at a time: four 16-byte table reads + three 8-byte ones in flight, not eight + six
                    const int e2 = (4 * mb + 2 * rp) / 2;
                    const row_d2 ma2 = row2[M_MIX / 2 + e2], pa2 = row2[(M_MIX + D) / 2 + e2];  // component 0 of the likelihood ...
                    const row_d2 mb2 = row2[(M_MIX + FUSED_MAX_COMPONENTS * 2 * D) / 2 + e2], pb2 = row2[(M_MIX + FUSED_MAX_COMPONENTS * 2 * D + D) / 2 + e2];  // ... and of the prior
                    const row_f2 lc2 = reinterpret_cast<const row_f2*>(locr)[e2], sc2 = reinterpret_cast<const row_f2*>(locr + D)[e2],
                                 rc2 = reinterpret_cast<const row_f2*>(locr + 2 * D)[e2];
#pragma unroll
                    for (int nb = 0; nb < 4; nb++) {
                        double xj[2];
#pragma unroll
                        for (int rr = 0; rr < 2; rr++) {  // (per accumulator: the same order of operations as before - rows ascending within nb)
                            const int r = 2 * rp + rr;
                            xj[rr] = (double)(T)acc[mb][nb][r];
                            const double ta = xj[rr] - ma2[rr], tb = xj[rr] - mb2[rr];
                            qpa[nb] = fma(ta * ta, pa2[rr], qpa[nb]);
                            qpb[nb] = fma(tb * tb, pb2[rr], qpb[nb]);
                        }
                        // the flow's input of the two rows on the packed fp32 instructions (flow_standardise2: same bits, 4 instructions per pair)
                        const flow_std2 z2 = flow_standardise2(flow_std2{(float)xj[0], (float)xj[1]}, lc2, sc2, rc2);
                        xf[(4 * mb + nb) * 4 + 2 * rp] = z2.x;
                        xf[(4 * mb + nb) * 4 + 2 * rp + 1] = z2.y;
                    }
                }
            // a particle's four lanes hold its partial sums: the same transpose brings them into the particle's own lane
            fused_transpose4(qpa[0], qpa[1], qpa[2], qpa[3]);
            fused_transpose4(qpb[0], qpb[1], qpb[2], qpb[3]);
            qa = (qpa[0] + qpa[1]) + (qpa[2] + qpa[3]);
            qb = (qpb[0] + qpb[1]) + (qpb[2] + qpb[3]);
            // mixture targets (<= FUSED_MAX_COMPONENTS components each): the further components' quadratic forms from the same
            // accumulators, folded into a running (max, sum) log-sum-exp in mixture_eval_regs' order (asmc_pcn.hip); a plain
            // Gaussian target (the headline) never enters the loops
            auto more_components = [&](int tt, int C, double q0c) -> double {
                double best = Lt[M_LOGW + tt * FUSED_MAX_COMPONENTS] - 0.5 * q0c;
                if (!MIX || C == 1) return best;
                double ssum = 1.0;
#pragma unroll 1
                for (int c = 1; c < C; c++) {
                    const double* __restrict__ tb = rowt + M_MIX + (tt * FUSED_MAX_COMPONENTS + c) * 2 * D;
                    double qp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int mb = 0; mb < 2; mb++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const double mc = tb[4 * mb + r], pc = tb[D + 4 * mb + r];
#pragma unroll
                            for (int nb = 0; nb < 4; nb++) {
                                const double tv = (double)(T)acc[mb][nb][r] - mc;
                                qp[nb] = fma(tv * tv, pc, qp[nb]);
                            }
                        }
                    fused_transpose4(qp[0], qp[1], qp[2], qp[3]);
                    const double tc = Lt[M_LOGW + tt * FUSED_MAX_COMPONENTS + c] - 0.5 * ((qp[0] + qp[1]) + (qp[2] + qp[3]));
                    if (tc > best) {
                        ssum = fma(ssum, exp(best - tc), 1.0);
                        best = tc;
                    } else if (tc > -INFINITY) {
                        ssum += exp(tc - best);
                    }
                }
                return best == -INFINITY ? -INFINITY : best + log(ssum);
            };
            mix_ll = more_components(0, p.c_ll, qa);
            mix_lp = more_components(1, p.c_lp, qb);
        } else {
#ifdef MV_COMPILER
#pragma unroll
        for (int g = 0; g < D / 4; g++) {
            const int j0 = 4 * g;
            double sr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < j0 + 4; k++) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (k <= j0 + r) sr[r] = fma(Lt[(j0 + r) * D + k], v[k], sr[r]);
                // ordering point every eight columns: the four running sums pass through an opaque statement that also
                // counts as a memory write, so this chunk's FMAs stay in front of it and the next chunk's coefficient
                // reads behind it.  (Left alone, instruction selection emits all ~560 LDS reads of the tile first and
                // the FMAs after them, and the reads' results spill.)
                if ((k & (MV_CHUNK - 1)) == MV_CHUNK - 1 || k == j0 + 3) asm volatile("" : "+v"(sr[0]), "+v"(sr[1]), "+v"(sr[2]), "+v"(sr[3])::"memory");
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = j0 + r;
                const double xj = (double)(T)(Lt[T_MU + j] + sr[r]);
                const double ta = xj - Lt[T_LLMU + j], tb = xj - Lt[T_LPMU + j];
                qa = fma(ta * ta, Lt[T_LLPR + j], qa);
                qb = fma(tb * tb, Lt[T_LPPR + j], qb);
                xf[j] = flow_standardise((float)xj, locs[j], locs[D + j], locs[2 * D + j]);
            }
            // ... and one behind the group's epilogue, so that its 28 table reads are not issued a row group early
            asm volatile("" : "+v"(qa), "+v"(qb), "+v"(xf[j0]), "+v"(xf[j0 + 1]), "+v"(xf[j0 + 2]), "+v"(xf[j0 + 3])::"memory");
        }
#else
        // The coefficient reads are issued by hand, three batches ahead of their FMAs.  A batch = the two coefficients
        // L[j0+r][2b], L[j0+r][2b+1] of the four rows of a row group (four 16-byte broadcast reads, one base register +
        // immediate offsets); the 72 batches of the lower triangle form ONE stream across the row groups, so the queue
        // never drains.  Left to hipcc the reads come two batches at a time with a full wait in front of every eight FMAs
        // (66 LDS latencies per tile: 16 k of the wave's 71 k cycles per tile when it runs alone).  The waits count only
        // this stream's own reads: LDS returns in order, so "at most 4 (P - 1) younger operations outstanding" implies the
        // batch is there whatever hipcc has put into the queue in between.  Same FMA order per row as before.
        {
            constexpr int NB = (D / 4) * (D / 4 + 1), P = MV_DEPTH;  // sum over g of (2 g + 2) column pairs
            const unsigned lbase = (unsigned)(size_t)tl + (unsigned)zoff;
            double2v ring[P][4];
            double sr[4] = {0.0, 0.0, 0.0, 0.0};
            auto issue = [&](auto bc) {
                constexpr int bi = decltype(bc)::value, g = mv_group(bi), b = bi - g * (g + 1), slot = bi % P;
                constexpr int o0 = ((4 * g + 0) * D + 2 * b) * 8, o1 = o0 + D * 8, o2 = o1 + D * 8, o3 = o2 + D * 8;
                auto& rg = ring[slot];  // (asm operands do not capture: name the captured objects first)
                const unsigned lb = lbase;
                asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"
                             "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                             : "=&v"(rg[0]), "=&v"(rg[1]), "=&v"(rg[2]), "=&v"(rg[3])
                             : "v"(lb), "n"(o0), "n"(o1), "n"(o2), "n"(o3));
            };
            auto consume = [&](auto bc) {
                constexpr int bi = decltype(bc)::value, g = mv_group(bi), b = bi - g * (g + 1), slot = bi % P, j0 = 4 * g;
                constexpr int ahead = (NB - 1 - bi) < (P - 1) ? (NB - 1 - bi) : (P - 1);
                if constexpr (bi + P - 1 < NB) issue(std::integral_constant<int, (bi + P - 1 < NB ? bi + P - 1 : 0)>{});
                auto& rg = ring[slot];
                static_assert(ahead <= 3, "lgkmcnt holds 15");
                if constexpr (ahead == 0)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]));
                else if constexpr (ahead == 1)
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]));
                else if constexpr (ahead == 2)
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]));
                else
                    asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(rg[0]), "+v"(rg[1]), "+v"(rg[2]), "+v"(rg[3]));
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (2 * b + e <= j0 + r) sr[r] = fma(rg[r][e], v[2 * b + e], sr[r]);
                if constexpr (b == 2 * g + 1) {  // the row group is complete
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int j = j0 + r;
                        const double xj = (double)(T)(Lt[T_MU + j] + sr[r]);
                        const double ta = xj - Lt[T_LLMU + j], tb = xj - Lt[T_LPMU + j];
                        qa = fma(ta * ta, Lt[T_LLPR + j], qa);
                        qb = fma(tb * tb, Lt[T_LPPR + j], qb);
                        xf[j] = flow_standardise((float)xj, locs[j], locs[D + j], locs[2 * D + j]);
                        sr[r] = 0.0;
                    }
                    // pinned behind the group's epilogue, so that its 28 table reads are not issued a row group early
                    auto &a0 = qa, &a1 = qb;
                    auto &f0 = xf[j0], &f1 = xf[j0 + 1], &f2 = xf[j0 + 2], &f3 = xf[j0 + 3];
                    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)::"memory");
                }
            };
            mv_for_each<0, (P - 1 < NB ? P - 1 : NB)>(issue);
            mv_for_each<0, NB>(consume);
        }
#endif
        }  // !MVM
        const double nll = MVM ? mix_ll : Lt[T_LOGW] - 0.5 * qa;
        const double nlp = MVM ? mix_lp : Lt[T_LOGW + 1] - 0.5 * qb;
        STAMP(3);
#if !FUSED_INPLACE
        if (valid) {
#pragma unroll
            for (int j = 0; j < D; j++) soa_store<T>(ypr, ys_lane, ys_tile + (unsigned)j * ys_row, v[j]);
        }
#endif
        // everything of the acceptance test that does not need log q(x') is finished HERE, and pinned: left to the
        // scheduler these computations sink below the flow, y' (64 VGPRs) stays alive across it for |y'|^2, and the
        // accumulators spill.  log_p_t(ll', lp', lq') = (1 - beta) lq' + beta (ll' + lp'): the second product is formed now.
        // (a +inf or NaN target term turns the folded constant into NaN: rejected, see log_p_t)
        double t2 = p.beta * (nll + nlp);
        t2 = (t2 < INFINITY) ? t2 : __builtin_nan("");
        double c1 = ref_corr(q1, p.nu, dn);
        double rhs = log_p_t(oll, olp, olq, p.beta) + ref_corr(q0, p.nu, dn);
        // (bm_log_unit: the noise generator's < 1 ulp log for positive normal arguments - the uniform is (k + 1/2) 2^-53 -
        // without libm's special-case and range code: a third of the instructions)
        double logu = bm_log_unit(accept_uniform(p.seed, gid, step));
        double kll = nll, klp = nlp;
        // ... folded into ONE number: accept  <=>  log u < ((1 - beta) lq' + t2 + c1) - rhs  <=>  (1 - beta) lq' + kacc > 0.  Four
        // doubles fewer alive across the flow (the kernel sits at the 256-register edge: they were spills); the re-association moves
        // a decision only where its margin is within an ulp or two of the sum (the razor edges the parity tests allow for)
        double kacc = ((t2 + c1) - rhs) - logu;
        asm volatile("" : "+v"(kacc), "+v"(kll), "+v"(klp));
#ifndef FUSED_NOSB
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- phase 2: two flow tiles (particles 0-31 and 32-63 of this wave) on the MFMA -----------------------------
        float lqt[2];
        unsigned tn_l = 0;
        float xaA[1][H / 2], xbA[1][H / 2], xaB[1][H / 2], xbB[1][H / 2];
        if constexpr (MVM) {
            // lane (g, n) holds rows 4 g + r (mb = 0) and 16 + 4 g + r (mb = 1) of particles 16 nb + n.  Flow tile A = particles 0-31
            // (nb = 0 in the even lane groups, nb = 1 in the odd ones), lane half hh = g >> 1: swapping the odd rows of the nb = 0
            // register with the even rows of the nb = 1 register hands every lane the four rows its neighbour group computed for
            // ITS particle - (first result: rows 8 hh + r, second: rows 8 hh + 4 + r).  Tile B likewise from nb = 2, 3.
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const auto sA = __builtin_amdgcn_permlane16_swap(__float_as_uint(xf[(4 * mb + 0) * 4 + r]), __float_as_uint(xf[(4 * mb + 1) * 4 + r]), false, false);
                    const auto sB = __builtin_amdgcn_permlane16_swap(__float_as_uint(xf[(4 * mb + 2) * 4 + r]), __float_as_uint(xf[(4 * mb + 3) * 4 + r]), false, false);
                    if (mb == 0) {
                        xaA[0][r] = __uint_as_float(sA[0]), xaA[0][4 + r] = __uint_as_float(sA[1]);
                        xaB[0][r] = __uint_as_float(sB[0]), xaB[0][4 + r] = __uint_as_float(sB[1]);
                    } else {
                        xbA[0][r] = __uint_as_float(sA[0]), xbA[0][4 + r] = __uint_as_float(sA[1]);
                        xbB[0][r] = __uint_as_float(sB[0]), xbB[0][4 + r] = __uint_as_float(sB[1]);
                    }
                }
        } else {
#pragma unroll
        for (int r = 0; r < H / 2; r++) {
            const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf[r]), __float_as_uint(xf[H / 2 + r]), false, false);
            xaA[0][r] = __uint_as_float(s1[0]);
            xaB[0][r] = __uint_as_float(s1[1]);
            const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf[H + r]), __float_as_uint(xf[H + H / 2 + r]), false, false);
            xbA[0][r] = __uint_as_float(s2[0]);
            xbB[0][r] = __uint_as_float(s2[1]);
        }
        }
        if constexpr (KIND == ASMC_FLOW_MAF) {
            // an autoregressive tile holds 16 coordinates per lane: the two coupling-style halves back to back (maf_coord, asmc_flow.hip)
            float xA[16], xB[16];
#pragma unroll
            for (int r = 0; r < 8; r++) xA[r] = xaA[0][r], xA[8 + r] = xbA[0][r], xB[r] = xaB[0][r], xB[8 + r] = xbB[0][r];
#ifndef FUSED_NOPRIO
            if (first_on_simd)
                __builtin_amdgcn_s_setprio(FUSED_PRIO_A);
            else
                __builtin_amdgcn_s_setprio(FUSED_PRIO_B);
#endif
            STAMP(4);
            tn_l = tile_fetch();  // the next tile's index: back long before the flow is through
            // (the affine form - asmc_coupling.affine, autoregressive flows - reaches the layers as a compile-time constant: a run-time
            // argument is a uniform branch per coordinate inside the epilogue; round 6)
            auto maf_tile = [&](float(&xv)[16], auto form_c) __attribute__((always_inline)) -> float {
                constexpr int FORM = decltype(form_c)::value;
                float ladj = 0.0f;
                unsigned amax_pk = 0u;
                for (int c = 0; c < n_layers; c++) {
                    float cond[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) cond[r] = xv[r];
                    coupling_layer_hs1p<HF, W, false, FORM, true>(cond, xv, sp + (size_t)c * FD::LAYER, lane, hh, ladj, amax_pk, FORM);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float amax = range_pk_max(amax_pk);
                float q = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; r++) q += xv[r] * xv[r];
                q += __shfl_xor(q, 32);
                const float lj = ladj + __shfl_xor(ladj, 32);
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                return !(amax < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
            };
            if (affine == 0) {  // (wave uniform)
                lqt[0] = maf_tile(xA, std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);  // one tile's accumulator chains at a time
                STAMP(5);
                lqt[1] = maf_tile(xB, std::integral_constant<int, 0>{});
            } else {
                lqt[0] = maf_tile(xA, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                STAMP(5);
                lqt[1] = maf_tile(xB, std::integral_constant<int, 1>{});
            }
        } else {
        // (two explicit calls: as a loop over the tiles the flow's A-operand reads become loop invariant and LLVM hoists
        // all 448 of them in front of it)
        auto flow_tile = [&](float(&xa)[1][H / 2], float(&xb)[1][H / 2]) __attribute__((always_inline)) -> float {
            float ladj[1] = {0.0f};
            float amax = 0.0f;  // largest operand the split-fp16 layers converted for this lane's particle
#ifndef FUSED_NOFLOW
            for (int c = 0; c < n_layers; c++) {
                const float* lpk = sp + (size_t)c * FD::LAYER;
                if (HS) {
                    if ((c & 1) == 0)
                        coupling_layer_hs<H, W>(xa[0], xb[0], lpk, lane, hh, ladj[0], amax);
                    else
                        coupling_layer_hs<H, W>(xb[0], xa[0], lpk, lane, hh, ladj[0], amax);
                } else if ((c & 1) == 0)
                    coupling_layer<H, W, 1>(xa, xb, lpk, lane, hh, ladj);
                else
                    coupling_layer<H, W, 1>(xb, xa, lpk, lane, hh, ladj);
            }
#endif
            float q = 0.0f;
#pragma unroll
            for (int r = 0; r < H / 2; r++) q += xa[0][r] * xa[0][r] + xb[0][r] * xb[0][r];
            q += __shfl_xor(q, 32);
            const float lj = ladj[0] + __shfl_xor(ladj[0], 32);
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            // an operand past the fp16 range: the pair was inf / NaN and the density is garbage - make it NaN
            return !(amax < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
        };
        // The two waves of a SIMD (waves w and w + 4 of the block) run the same program; left alone they fall into
        // lockstep - both in the vector phases, then both fighting for the matrix pipe - and the MFMA sits idle half the
        // time.  Static priorities break the tie: a wave in its matrix phase outranks a wave in a vector phase, and the
        // first wave of a SIMD outranks the second when both are in the matrix phase, which pushes them into anti-phase.
#ifndef FUSED_NOPRIO
        if (first_on_simd)
            __builtin_amdgcn_s_setprio(FUSED_PRIO_A);
        else
            __builtin_amdgcn_s_setprio(FUSED_PRIO_B);
#endif
        STAMP(4);
        tn_l = tile_fetch();  // the next tile's index: back long before the flow is through
#ifndef FUSED_FLOW2
#define FUSED_FLOW2 1  // both tiles through each coupling layer together, MFMAs and conversions interleaved by hand (asmc_flow_dev.h)
#endif
        if (HS && FUSED_INPLACE) {  // one tile at a time, conversions in the shadow of the tile's own MFMAs; y' stays in registers
            auto one_tile = [&](float(&xa)[1][H / 2], float(&xb)[1][H / 2]) __attribute__((always_inline)) -> float {
                float ladj = 0.0f;
                unsigned amax_pk = 0u;
                for (int c = 0; c < n_layers; c++) {
                    const float* lpk = sp + (size_t)c * FD::LAYER;
                    if ((c & 1) == 0)
                        coupling_layer_hs1p<H, W, HS1P_PREFETCH, 0, true>(xa[0], xb[0], lpk, lane, hh, ladj, amax_pk);
                    else
                        coupling_layer_hs1p<H, W, HS1P_PREFETCH, 0, true>(xb[0], xa[0], lpk, lane, hh, ladj, amax_pk);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float amax = range_pk_max(amax_pk);
                float q = 0.0f;
#pragma unroll
                for (int r = 0; r < H / 2; r++) q += xa[0][r] * xa[0][r] + xb[0][r] * xb[0][r];
                q += __shfl_xor(q, 32);
                const float lj = ladj + __shfl_xor(ladj, 32);
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                return !(amax < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
            };
            lqt[0] = one_tile(xaA, xbA);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(5);
            lqt[1] = one_tile(xaB, xbB);
        } else if (HS && FUSED_FLOW2 && W < 128) {  // (W = 128: the two interleaved tiles need 288 accumulator registers - one tile at a time there)
            float ladjA = 0.0f, ladjB = 0.0f;
            unsigned amaxA = 0u, amaxB = 0u;  // packed fp16 running maxima of the operands' hi halves
            for (int c = 0; c < n_layers; c++) {
                const float* lpk = sp + (size_t)c * FD::LAYER;
                if ((c & 1) == 0)
                    coupling_layer_hs2<H, W>(xaA[0], xbA[0], xaB[0], xbB[0], lpk, lane, hh, ladjA, ladjB, amaxA, amaxB);
                else
                    coupling_layer_hs2<H, W>(xbA[0], xaA[0], xbB[0], xaB[0], lpk, lane, hh, ladjA, ladjB, amaxA, amaxB);
                __builtin_amdgcn_sched_barrier(0);
            }
            auto finish = [&](const float(&xa)[1][H / 2], const float(&xb)[1][H / 2], float ladj, unsigned amax_pk) -> float {
                float amax = range_pk_max(amax_pk);
                float q = 0.0f;
#pragma unroll
                for (int r = 0; r < H / 2; r++) q += xa[0][r] * xa[0][r] + xb[0][r] * xb[0][r];
                q += __shfl_xor(q, 32);
                const float lj = ladj + __shfl_xor(ladj, 32);
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                return !(amax < FLOW_HS_MAX) ? __builtin_nanf("") : (-0.5f * q + base_const) + (ladj0 + lj);
            };
            lqt[0] = finish(xaA, xbA, ladjA, amaxA);
            lqt[1] = finish(xaB, xbB, ladjB, amaxB);
        } else {
            lqt[0] = flow_tile(xaA, xbA);
            __builtin_amdgcn_sched_barrier(0);  // one tile's accumulator chains at a time
            STAMP(5);
            lqt[1] = flow_tile(xaB, xbB);
        }
        }  // KIND
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        STAMP(6);
        const double nlq = (double)(hh == 0 ? lqt[0] : lqt[1]);
        // ---- phase 3: accept, on the lane's own particle ---------------------------------------------------------------
        // a non-finite log q(x') (NaN, or +-inf: fp16 operand overflow at |activation| >= 65504 in the split products,
        // asmc_flow_dev.h) rejects the proposal; counted, so that the host can tell the user
        const bool lq_finite = fabs(nlq) < INFINITY;
        if (valid && !lq_finite) n_bad++;
        const bool accepted = valid && lq_finite && fma(1.0 - p.beta, nlq, kacc) > 0.0;  // (a NaN density compares false: rejected)
#if FUSED_INPLACE
        // y' is stored FIRST: its registers are free once the stores have issued, so the next tile's state loads into them - with
        // the loads in front (round 3's order, when y' was parked and dead by now) both tiles' rows are alive at once and the loop
        // ends with 64 register copies behind a full wait for the loads (2-10 k cycles per tile on the stamps)
        if (accepted) {
            ll[i] = kll;
            lp[i] = klp;
            lq[i] = nlq;
            n_acc++;
        }
        if constexpr (MVM) {
            // y' sits transposed: register 4 s + r = (coordinate 4 s + g, particle 16 r + n) in lane (g, n) - still runs of 16
            // consecutive particles of one coordinate.  Register r's lanes are live where particle 16 r + n accepted.
            const unsigned long long accmask = __ballot(accepted);
            const unsigned ys_lane_t = (unsigned)(lane >> 4) * ys_row + (unsigned)(lane & 15) * (unsigned)sizeof(T);
            const __amdgpu_buffer_rsrc_t ysw = __builtin_amdgcn_make_buffer_rsrc(ysu, 0, half_records, 0x00020000);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if ((accmask >> (16 * r + (lane & 15))) & 1ull) {
#pragma unroll
                    for (int sI = 0; sI < D / 4; sI++)
                        soa_store<T>(ysw, ys_lane_t, ys_tile + (unsigned)(4 * sI) * ys_row + (unsigned)(16 * r) * (unsigned)sizeof(T), v[4 * sI + r]);
                }
            }
        } else if (accepted) {  // y' is still in registers: the accepted lanes' rows change in place, nothing else moves
            const __amdgpu_buffer_rsrc_t ysw = __builtin_amdgcn_make_buffer_rsrc(ysu, 0, half_records, 0x00020000);
#pragma unroll
            for (int j = 0; j < D; j++) soa_store<T>(ysw, ys_lane, ys_tile + (unsigned)j * ys_row, v[j]);
        }
        // ... and the next tile's state goes into flight behind them
        const unsigned tn = (unsigned)__builtin_amdgcn_readfirstlane((int)tn_l);
        const bool have_n = (int64_t)tn < n_tiles;
        unsigned par_n = 0;
        double vn[D];
#pragma unroll
        for (int j = 0; j < D; j++) vn[j] = 0.0;  // (left undefined for the last tile, LLVM carries the OLD y through the whole body instead: 64 VGPRs, spills)
        double olln = 0.0, olpn = 0.0, olqn = 0.0;
        if (have_n) {
            par_n = tile_parity(tn);
            tile_load(tn, par_n, vn, olln, olpn, olqn);
        }
#else
        // the next tile's state goes into flight in front of this tile's stores and copies
        const unsigned tn = (unsigned)__builtin_amdgcn_readfirstlane((int)tn_l);
        const bool have_n = (int64_t)tn < n_tiles;
        unsigned par_n = 0;
        double vn[D];
#pragma unroll
        for (int j = 0; j < D; j++) vn[j] = 0.0;  // (left undefined for the last tile, LLVM carries the OLD y through the whole body instead: 64 VGPRs, spills)
        double olln = 0.0, olpn = 0.0, olqn = 0.0;
        if (have_n) {
            par_n = tile_parity(tn);
            tile_load(tn, par_n, vn, olln, olpn, olqn);
        }
        if (accepted) {
            ll[i] = kll;
            lp[i] = klp;
            lq[i] = nlq;
            n_acc++;
        }
        {
            const int n_a = __builtin_popcountll(__ballot(accepted)), n_r = __builtin_popcountll(__ballot(valid && !accepted));
            const bool flip = n_a > 0 && n_r <= n_a;  // wave uniform: the state moves to half B
            if (flip ? n_r > 0 : n_a > 0) {           // somebody has to be copied (src -> dst)
                const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(flip == (par != 0) ? ysu1 : ysu, 0, half_records, 0x00020000);
                const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(flip == (par != 0) ? ysu : ysu1, 0, half_records, 0x00020000);
                if (valid && accepted != flip) {
                    double w[D];
#pragma unroll
                    for (int j = 0; j < D; j++) w[j] = soa_load<T>(src, ys_lane, ys_tile + (unsigned)j * ys_row);
#pragma unroll
                    for (int j = 0; j < D; j++) soa_store<T>(dst, ys_lane, ys_tile + (unsigned)j * ys_row, w[j]);
                }
            }
            if (flip && lane == 0) tile_par[t] = (unsigned char)(par ^ 1u);
        }
#endif
        STAMP(7);
#ifdef FUSED_STAMP
        ntile++;
#endif
#ifndef FUSED_NOSB
        __builtin_amdgcn_sched_barrier(0);
#endif
        t = tn, par = par_n, have = have_n;
        oll = olln, olp = olpn, olq = olqn;
#pragma unroll
        for (int j = 0; j < D; j++) v[j] = vn[j];
    }
#ifdef FUSED_STAMP
    if (lane == 0 && blockIdx.x == 7 && (wave == 0 || wave == 4) && ntile > 0)
        printf("wave %d tiles %d | fetch %llu yload %llu noise %llu matvec %llu park+swap %llu flowA %llu flowB %llu accept %llu (cycles per tile)\n", wave, ntile,
               tsum[0] / ntile, tsum[1] / ntile, tsum[2] / ntile, tsum[3] / ntile, tsum[4] / ntile, tsum[5] / ntile, tsum[6] / ntile, tsum[7] / ntile);
#endif
    __shared__ long long s_cnt[THREADS / 64];
    n_acc = wave_sum_ll(n_acc);
    n_bad = wave_sum_ll(n_bad);
    if (lane == 0 && n_bad != 0 && ad.nonfinite) atomicAdd(ad.nonfinite, (unsigned long long)n_bad);
    if (lane == 0) s_cnt[wave] = n_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tsum = 0;
        for (int w = 0; w < THREADS / 64; w++) tsum += s_cnt[w];
        block_counts[blockIdx.x] = tsum;
    }
    if (ad.done == nullptr) return;
    // the block that finishes last closes the step: total accept count, step-size adaptation (k_pcn_adapt's arithmetic)
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned int tk = __hip_atomic_fetch_add(ad.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == gridDim.x - 1u);
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    long long c = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += THREADS)
        c += __hip_atomic_load(&block_counts[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c = wave_sum_ll(c);
    __syncthreads();
    if (lane == 0) s_cnt[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = 0;
        for (int w = 0; w < THREADS / 64; w++) c += s_cnt[w];
        if (ad.cell) {  // sharded: hand the rank's count to the exchange hook (lagged: the step's own cell of the block)
            ad.cell[ad.adapt >= 2 ? ad.t % ad.adapt : 0] = c;
            return;
        }
        ad.counts_out[ad.t] = c;
        ad.rho_hist[ad.t] = rho;
        if (ad.adapt == 1) {
            *ad.rho = pcn_adapt_rho(rho, c, ad.n, ad.target, ad.t);
        } else if (ad.adapt >= 2 && ((ad.t + 1) % ad.adapt == 0 || ad.last)) {  // lagged: the block's updates, in order
            double r = rho;
            for (int tp = ad.t - (ad.t % ad.adapt); tp <= ad.t; tp++)
                r = pcn_adapt_rho(r, tp == ad.t ? c : ad.counts_out[tp], ad.n, ad.target, tp);
            *ad.rho = r;
        }
    }
}

template <typename T>
static int launch_pcn_flow_fused(asmc_ctx* ctx, int64_t n, double* ll, double* lp, double* lq, const PcnDev& pd,
                                 const asmc_coupling* f, const double* rho_ptr, uint32_t step, unsigned int* tile_counter,
                                 long long* block_counts, int* grid_out, const PcnAdaptArgs& adapt, hipStream_t st) {
    PcnScalars ps;
    ps.beta = pd.beta;
    ps.nu = pd.nu;
    ps.gam = pd.gam;
    ps.ys = pd.ys;
    ps.n_pad = pd.n_pad;
    ps.d_real = pd.d;
    ps.d_noise = pd.d_noise;
    ps.seed = pd.seed;
    ps.gid0 = pd.gid0;
    ps.c_ll = pd.ll.C;
    ps.c_lp = pd.lp.C;
    ps.c_lq = 0;
    ps.bmtab = pd.bmtab;
    ps.tile_par = pd.tile_par;
    const float ladj0 = (float)(-f->log_scale_sum);
    const float base_const = (float)(-0.5 * f->dims * 1.8378770664093453);
    const int64_t n_tiles = (n + 63) / 64;
    const int grid = (int)(n_tiles < (int64_t)ctx->num_cu * (FUSED_THREADS / 64) ? (n_tiles + FUSED_THREADS / 64 - 1) / (FUSED_THREADS / 64) : ctx->num_cu);  // one 8-wave block per CU
    *grid_out = grid;
    // flow arithmetic: split-fp16 MFMA (fp32-equivalent operands, asmc_flow_dev.h) unless ASMC_FLOW_MATH=f32 asks for the
    // fp32 MFMA chain
    // (Hidden width 128 goes through its coupling layers ONE flow tile at a time: the two interleaved tiles of coupling_layer_hs2
    // need 288 accumulator registers there, the compiler spilled ~300 around the hand-scheduled conversion / MFMA sequence and
    // the step returned run-to-run different log q - tools/stress_fused.py.)
    const bool hs = asmc_flow_math_split();
    const bool mix = pd.ll.C > 1 || pd.lp.C > 1;
#define ASMC_FUSED_CASE_K(WW, NZ, HSV, KD) ASMC_FUSED_CASE_KM(WW, NZ, HSV, KD, false)
#define ASMC_FUSED_CASE_KM(WW, NZ, HSV, KD, MX)                                                                           \
    if (f->hidden == WW && pd.noise == NZ && hs == HSV && f->kind == KD && mix == MX) {                                   \
        auto kern = k_pcn_flow_fused<T, WW, NZ, HSV, KD, MX>;                                                                 \
        const size_t lds0 = (size_t)f->n_layers * FlowDims<(KD == ASMC_FLOW_MAF ? 32 : 16), WW>::LAYER * sizeof(float) + FUSED_TL_DOUBLES * sizeof(double) + BM_TAB_N * sizeof(bm_d2); \
        const size_t par_bytes = (size_t)((n_tiles + 31) / 32) * 4;   /* the tiles' parity bits ride in LDS when they fit */ \
        const int par_words = (!FUSED_INPLACE && lds0 + par_bytes <= 160 * 1024) ? (int)(par_bytes / 4) : 0;                     \
        const size_t lds = lds0 + (size_t)par_words * 4;                                                                     \
        static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];                                                                                      \
        if (lds > 64 * 1024 && lds > attr_lds) {                                                                         \
            ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_lds = lds;                                                                                              \
        }                                                                                                                \
        ASMC_LAUNCH(ctx, st, "k_pcn_flow_fused", kern, dim3(grid), dim3(FUSED_THREADS), lds, st, n, ll, lp, lq,                     \
                    (const double*)ctx->d_ptab, ps, rho_ptr, step, f->packed_dev, (int)f->n_layers, f->loc_dev, f->scale_dev, \
                    ladj0, base_const, tile_counter, block_counts, adapt, par_words, KD == ASMC_FLOW_MAF ? (int)f->affine : 0);  \
        ASMC_LAUNCH_CHECK();                                                                                             \
        return ASMC_OK;                                                                                                  \
    }
#define ASMC_FUSED_CASE(WW, NZ, HSV) ASMC_FUSED_CASE_K(WW, NZ, HSV, ASMC_FLOW_COUPLING)
    ASMC_FUSED_CASE(64, ASMC_NOISE_F64, true)
#ifndef FUSED_ONLY_HEADLINE
    ASMC_FUSED_CASE_KM(64, ASMC_NOISE_F64, true, ASMC_FLOW_COUPLING, true)  // mixture targets
    ASMC_FUSED_CASE_KM(64, ASMC_NOISE_F32, true, ASMC_FLOW_COUPLING, true)  // (diagnostic builds compile the headline instantiation alone: seconds instead of minutes)
    ASMC_FUSED_CASE(64, ASMC_NOISE_F64, false)
    ASMC_FUSED_CASE(64, ASMC_NOISE_F32, true)
    ASMC_FUSED_CASE(64, ASMC_NOISE_F32, false)
    ASMC_FUSED_CASE(32, ASMC_NOISE_F64, true)
    ASMC_FUSED_CASE(32, ASMC_NOISE_F64, false)
    ASMC_FUSED_CASE(32, ASMC_NOISE_F32, true)
    ASMC_FUSED_CASE(32, ASMC_NOISE_F32, false)
    ASMC_FUSED_CASE(128, ASMC_NOISE_F64, true)
    ASMC_FUSED_CASE(128, ASMC_NOISE_F64, false)
    ASMC_FUSED_CASE(128, ASMC_NOISE_F32, true)
    ASMC_FUSED_CASE(128, ASMC_NOISE_F32, false)
    // masked autoregressive transforms (split-fp16 layers; the fp32 MFMA chain takes the split propose / flow / accept kernels)
    ASMC_FUSED_CASE_K(64, ASMC_NOISE_F64, true, ASMC_FLOW_MAF)
    ASMC_FUSED_CASE_K(64, ASMC_NOISE_F32, true, ASMC_FLOW_MAF)
    ASMC_FUSED_CASE_K(32, ASMC_NOISE_F64, true, ASMC_FLOW_MAF)
    ASMC_FUSED_CASE_K(32, ASMC_NOISE_F32, true, ASMC_FLOW_MAF)
#endif
#undef ASMC_FUSED_CASE
#undef ASMC_FUSED_CASE_K
#undef ASMC_FUSED_CASE_KM
    asmc_set_error("fused flow step: unsupported flow (kind %d, hidden width %d)", (int)f->kind, (int)f->hidden);
    return ASMC_ERR_UNSUPPORTED;
}

// whether the fused step covers this mutation: d = 32 on the coordinate-major whitened state, single-Gaussian targets,
// every coupling layer resident in LDS next to nothing else
bool asmc_pcn_flow_fused_ok(const asmc_pcn_params* prm, const asmc_coupling* f) {
    if (getenv("ASMC_FLOW_SPLIT")) return false;
    // d = 32 as it is; fewer dimensions zero-padded to 32 by asmc_pcn_mutate_flow (prm->d is 32 by then, the flow keeps its own)
    if (prm->d != 32 || f->dims > 32 || f->dims < 2) return false;
    if (f->dims != 32 && !(FUSED_MVMFMA && FUSED_INPLACE)) return false;  // (the padded layout lives in the matrix-core variant's tables)
    if (f->kind == ASMC_FLOW_COUPLING && (f->dims % 2)) return false;
    // mixture targets: the matrix-core variant takes up to FUSED_MAX_COMPONENTS components each
    const bool mvm = FUSED_MVMFMA && FUSED_INPLACE && f->kind == ASMC_FLOW_COUPLING && f->hidden == 64 && asmc_flow_math_split();  // (the mixture instantiations)
    const int cmax = mvm ? FUSED_MAX_COMPONENTS : 1;
    if (prm->log_likelihood.n_components < 1 || prm->log_likelihood.n_components > cmax) return false;
    if (prm->log_prior.n_components < 1 || prm->log_prior.n_components > cmax) return false;
    if (!(f->hidden == 32 || f->hidden == 64 || f->hidden == 128)) return false;
    if (f->kind == ASMC_FLOW_MAF) {  // split-fp16 layers only, widths whose accumulators fit one tile at a time
        if (!asmc_flow_math_split() || f->hidden == 128) return false;
        const size_t per_tr = (size_t)((2 * (f->hidden / 32) + 2) * 32 + f->hidden * 32 + f->hidden * f->hidden + 2 * 32 * f->hidden) * sizeof(float);
        return per_tr * (size_t)f->n_layers + FUSED_TL_DOUBLES * sizeof(double) + BM_TAB_N * sizeof(bm_d2) <= 160 * 1024;
    }
    const size_t per_layer = (size_t)((2 * (f->hidden / 32) + 1) * 32 + f->hidden * 16 + f->hidden * f->hidden + 2 * 16 * f->hidden) * sizeof(float);
    return per_layer * (size_t)f->n_layers + FUSED_TL_DOUBLES * sizeof(double) + BM_TAB_N * sizeof(bm_d2) <= 160 * 1024;
}


int asmc_pcn_flow_fused_launch(asmc_ctx* ctx, int64_t n, int x_dtype, double* ll, double* lp, double* lq, const PcnDev& pd,
                               const asmc_coupling* f, const double* rho_ptr, uint32_t step, unsigned int* tile_counter,
                               long long* block_counts, int* grid_out, const PcnAdaptArgs& adapt, hipStream_t st) {
    if (x_dtype == ASMC_F64)
        return launch_pcn_flow_fused<double>(ctx, n, ll, lp, lq, pd, f, rho_ptr, step, tile_counter, block_counts, grid_out, adapt, st);
    return launch_pcn_flow_fused<float>(ctx, n, ll, lp, lq, pd, f, rho_ptr, step, tile_counter, block_counts, grid_out, adapt, st);
}
