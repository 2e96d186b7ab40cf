// asmc_weights.hip — tempered log-weights, stable log-sum-exp / ESS reductions, evidence moments.
//
// Replaces (reference mj-will/aspire, paths relative to its root):
//   src/aspire/samples.py:1221-1224  SMCSamples.unnormalized_log_weights
//   src/aspire/utils.py:248-255      logsumexp   (max pass, then sum of exp(x - max))
//   src/aspire/utils.py:510-512      effective_sample_size
//   src/aspire/samples.py:1226-1249  log_evidence_ratio(_variance), log_weights
//
// Layout: ll, lp, lq are fp64 vectors [N] in HBM; one coalesced pass serves K candidate betas
// (the k-ary bisection of determine_beta, smc/base.py:177-185).  24 B per particle per pass.
// Reductions: per-lane accumulators -> wave64 __shfl_xor tree -> LDS combine of the 4 waves ->
// per-block partial in HBM -> fixed-order finalize kernel (bitwise reproducible, no fp atomics).
// Max pass uses integer atomicMax on an order-preserving key (order independent => deterministic).
// Compiled with -ffp-contract=off so lw is the reference's plain IEEE mul/mul/add sequence.
#include <stdlib.h>

#include "asmc_common.h"
#include "asmc_bisect.h"

template <int KT>
struct BetaPack {
    double c1[KT];     // beta0 - beta_k
    double c2[KT];     // beta_k - beta0
    double m[KT];      // max (when supplied by the host)
    double shift[KT];  // additive shift applied before subtracting m
};

// self-resetting arrival counters in ctx->d_bar (zeroed at creation), behind the persistent kernel's barrier counters
#define SHARD_TICKET_CELL (1024 * 12)
#define COUNT_CELLS (1024 * 12 + 64)  // two (NaN, inf) count slots of k_count_nonfinite, 8-byte aligned

__device__ __forceinline__ double lw_of(double ll, double lp, double lq, double c1, double c2) {
    // (self.beta - beta) * log_q + (beta - self.beta) * (log_likelihood + log_prior)
    double t1 = c1 * lq;
    double t2 = c2 * (ll + lp);
    return t1 + t2;
}

// ---------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_max(int64_t n, const double* __restrict__ ll,
                                                           const double* __restrict__ lp,
                                                           const double* __restrict__ lq,
                                                           BetaPack<KT> bp_arg,
                                                           unsigned long long* __restrict__ keys,
                                                           unsigned long long* __restrict__ nan_count,
                                                           const BetaPack<KT>* __restrict__ bp_dev,
                                                           const double* __restrict__ skip_flag) {
    if (skip_flag && *skip_flag != 0.0) return;  // device-side bisection already converged
    const BetaPack<KT>& bp = bp_dev ? *bp_dev : bp_arg;
    double mx[KT];
#pragma unroll
    for (int k = 0; k < KT; k++) mx[k] = -INFINITY;
    long long nn = 0;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double a = ll[i], b = lp[i], q = lq[i];
#pragma unroll
        for (int k = 0; k < KT; k++) {
            double lw = lw_of(a, b, q, bp.c1[k], bp.c2[k]);
            if (lw != lw)
                nn++;
            else
                mx[k] = fmax(mx[k], lw);
        }
    }
    __shared__ double s_mx[ASMC_BLOCK / 64][KT];
    __shared__ long long s_nn[ASMC_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KT; k++) {
        double v = wave_max(mx[k]);
        if (lane == 0) s_mx[wave][k] = v;
    }
    nn = wave_sum_ll(nn);
    if (lane == 0) s_nn[wave] = nn;
    __syncthreads();
    if (threadIdx.x < KT) {
        double v = s_mx[0][threadIdx.x];
        for (int w = 1; w < ASMC_BLOCK / 64; w++) v = fmax(v, s_mx[w][threadIdx.x]);
        atomicMax(&keys[threadIdx.x], f64_to_key(v));
    }
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < ASMC_BLOCK / 64; w++) t += s_nn[w];
        if (t) atomicAdd(nan_count, (unsigned long long)t);
    }
}

// S1_k = sum exp(t), S2_k = sum exp(t)^2, t = (lw + shift_k) - m_k
template <int KT>
__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_sums(int64_t n, const double* __restrict__ ll,
                                                            const double* __restrict__ lp,
                                                            const double* __restrict__ lq,
                                                            BetaPack<KT> bp_arg,
                                                            const unsigned long long* __restrict__ keys,
                                                            double* __restrict__ partials,
                                                            const BetaPack<KT>* __restrict__ bp_dev,
                                                            const double* __restrict__ skip_flag) {
    if (skip_flag && *skip_flag != 0.0) return;
    const BetaPack<KT>& bp = bp_dev ? *bp_dev : bp_arg;
    double s1[KT], s2[KT], m[KT];
#pragma unroll
    for (int k = 0; k < KT; k++) {
        s1[k] = 0.0;
        s2[k] = 0.0;
        m[k] = keys ? key_to_f64(keys[k]) : bp.m[k];
    }
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double a = ll[i], b = lp[i], q = lq[i];
#pragma unroll
        for (int k = 0; k < KT; k++) {
            double lw = lw_of(a, b, q, bp.c1[k], bp.c2[k]);
            double t = (lw + bp.shift[k]) - m[k];
            double e = exp(t);
            s1[k] += e;
            s2[k] += e * e;
        }
    }
    __shared__ double s_p[ASMC_BLOCK / 64][KT * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KT; k++) {
        double v1 = wave_sum(s1[k]);
        double v2 = wave_sum(s2[k]);
        if (lane == 0) {
            s_p[wave][2 * k] = v1;
            s_p[wave][2 * k + 1] = v2;
        }
    }
    __syncthreads();
    if (threadIdx.x < KT * 2) {
        double v = s_p[0][threadIdx.x];
        for (int w = 1; w < ASMC_BLOCK / 64; w++) v += s_p[w][threadIdx.x];
        partials[(size_t)blockIdx.x * (KT * 2) + threadIdx.x] = v;
    }
}

// one 64-lane block per output column: fixed-order reduction of the block partials
__global__ __launch_bounds__(64) void k_finalize_columns(int nblocks, int ncols,
                                                        const double* __restrict__ partials,
                                                        double* __restrict__ out, int out_stride,
                                                        int out_offset_mode,
                                                        const unsigned long long* __restrict__ keys,
                                                        const unsigned long long* __restrict__ nan_count) {
    const int col = blockIdx.x;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) v += partials[(size_t)b * ncols + col];
    v = wave_sum(v);
    if (threadIdx.x == 0) {
        if (out_offset_mode == 1) {
            // stats layout {m, S1, S2, n_nan} per beta; col = 2k + which
            const int k = col >> 1, which = col & 1;
            out[k * 4 + 1 + which] = v;
            if (which == 0) {
                out[k * 4 + 0] = key_to_f64(keys[k]);
                out[k * 4 + 3] = (double)(*nan_count);
            }
        } else {
            out[(size_t)col * out_stride] = v;
        }
    }
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_m2(int64_t n, const double* __restrict__ ll,
                                                          const double* __restrict__ lp,
                                                          const double* __restrict__ lq, double c1,
                                                          double c2, double m, double mean_u,
                                                          double* __restrict__ partials) {
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        double lw = lw_of(ll[i], lp[i], lq[i], c1, c2);
        double dlt = exp(lw - m) - mean_u;
        acc += dlt * dlt;
    }
    __shared__ double s_p[ASMC_BLOCK / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = s_p[0];
        for (int w = 1; w < ASMC_BLOCK / 64; w++) v += s_p[w];
        partials[blockIdx.x] = v;
    }
}

// k_weights_m2 plus the sum of the second log-sum-exp of the resampling step in the same pass:
// column 0: sum (exp(lw - m) - mean_u)^2 (samples.py:1230-1242); column 1: sum exp((lw + shift) - mp), the
// logsumexp of the SHIFTED log-weights (samples.py:1277 on samples.py:1244-1249)
__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_m2_lse(int64_t n, const double* __restrict__ ll,
                                                              const double* __restrict__ lp,
                                                              const double* __restrict__ lq, double c1, double c2,
                                                              double m, double mean_u, double shift, double mp,
                                                              double* __restrict__ partials) {
    double acc = 0.0, s1p = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double lw = lw_of(ll[i], lp[i], lq[i], c1, c2);
        const double dlt = exp(lw - m) - mean_u;
        acc += dlt * dlt;
        s1p += exp((lw + shift) - mp);
    }
    __shared__ double s_p[ASMC_BLOCK / 64][2];
    acc = wave_sum(acc);
    s1p = wave_sum(s1p);
    if ((threadIdx.x & 63) == 0) {
        s_p[threadIdx.x >> 6][0] = acc;
        s_p[threadIdx.x >> 6][1] = s1p;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = s_p[0][threadIdx.x];
        for (int w = 1; w < ASMC_BLOCK / 64; w++) v += s_p[w][threadIdx.x];
        partials[(size_t)blockIdx.x * 2 + threadIdx.x] = v;
    }
}

// mode 0: out = lw + shift           (SMCSamples.log_weights)
// mode 1: out = exp((lw + shift) - lse)   (normalised weights for resampling)
template <int MODE>
__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_map(int64_t n, const double* __restrict__ ll,
                                                           const double* __restrict__ lp,
                                                           const double* __restrict__ lq, double c1,
                                                           double c2, double shift, double lse,
                                                           double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        double lw = lw_of(ll[i], lp[i], lq[i], c1, c2) + shift;
        out[i] = MODE == 0 ? lw : exp(lw - lse);
    }
}

// NaN / inf census of v into counters[0..1].  No memset in front of it: the counters are TWO slots used in turn - a launch adds
// into the slot the previous launch zeroed (the first one: zeroed at creation) and zeroes the other one for the next launch
// (`next_slot`: nobody else touches it during this launch).  Atomics only where something was found.
__global__ __launch_bounds__(ASMC_BLOCK) void k_count_nonfinite(int64_t n, const double* __restrict__ v,
                                                               unsigned long long* __restrict__ counters,
                                                               unsigned long long* __restrict__ next_slot) {
    if (blockIdx.x == 0 && threadIdx.x < 2) next_slot[threadIdx.x] = 0ull;
    long long n_nan = 0, n_inf = 0;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        double x = v[i];
        if (x != x)
            n_nan++;
        else if (isinf(x))
            n_inf++;
    }
    n_nan = wave_sum_ll(n_nan);
    n_inf = wave_sum_ll(n_inf);
    if ((threadIdx.x & 63) == 0) {
        if (n_nan) atomicAdd(&counters[0], (unsigned long long)n_nan);
        if (n_inf) atomicAdd(&counters[1], (unsigned long long)n_inf);
    }
}

// ---------------------------------------------------------------------------------------------
template <int KT>
static void fill_pack(BetaPack<KT>& bp, double beta0, const double* betas, const double* m,
                      const double* shift, int K) {
    for (int k = 0; k < KT; k++) {
        int kk = k < K ? k : K - 1;  // pad with the last candidate
        bp.c1[k] = beta0 - betas[kk];
        bp.c2[k] = betas[kk] - beta0;
        bp.m[k] = m ? m[kk] : 0.0;
        bp.shift[k] = shift ? shift[kk] : 0.0;
    }
}

static int bucket_of(int K) {
    int kt = 1;
    while (kt < K) kt <<= 1;
    return kt;
}

#define ASMC_DISPATCH_KT(KTV, BODY) \
    switch (KTV) {                  \
        case 1: { constexpr int KT = 1; BODY; } break;   \
        case 2: { constexpr int KT = 2; BODY; } break;   \
        case 4: { constexpr int KT = 4; BODY; } break;   \
        case 8: { constexpr int KT = 8; BODY; } break;   \
        case 16: { constexpr int KT = 16; BODY; } break; \
        default: { constexpr int KT = 32; BODY; } break; \
    }

static int reduce_grid(const asmc_ctx* ctx, int64_t n, int kt) {
    // enough work per thread to amortise the K-way block reduction; cap by scratch size
    int per_block = ASMC_BLOCK * (kt >= 16 ? 4 : 8);
    int cap = ctx->num_cu * 4;
    if (cap > ASMC_MAX_BLOCKS) cap = ASMC_MAX_BLOCKS;
    return grid_for(n, per_block, cap);
}

static int launch_max(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                      double beta0, const double* betas, int K, hipStream_t st) {
    ASMC_HIP(hipMemsetAsync(ctx->d_keys, 0, sizeof(unsigned long long) * (ASMC_MAX_BETAS + 8), st));
    const int kt = bucket_of(K);
    const int grid = reduce_grid(ctx, n, kt);
    ASMC_DISPATCH_KT(kt, {
        BetaPack<KT> bp;
        fill_pack<KT>(bp, beta0, betas, nullptr, nullptr, K);
        ASMC_LAUNCH(ctx, st, "k_weights_max<KT>", k_weights_max<KT>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, bp,
                    ctx->d_keys, ctx->d_keys + ASMC_MAX_BETAS, (const BetaPack<KT>*)nullptr, (const double*)nullptr);
    });
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static int launch_sums(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                       double beta0, const double* betas, const double* m, const double* shift, int K,
                       bool m_from_keys, int* grid_out, hipStream_t st) {
    const int kt = bucket_of(K);
    const int grid = reduce_grid(ctx, n, kt);
    *grid_out = grid;
    ASMC_DISPATCH_KT(kt, {
        BetaPack<KT> bp;
        fill_pack<KT>(bp, beta0, betas, m, shift, K);
        ASMC_LAUNCH(ctx, st, "k_weights_sums<KT>", k_weights_sums<KT>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, bp,
                    m_from_keys ? ctx->d_keys : nullptr, ctx->d_partials, (const BetaPack<KT>*)nullptr, (const double*)nullptr);
    });
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static int check_common(asmc_ctx* ctx, int64_t n, const void* a, const void* b, const void* c) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    ASMC_REQUIRE(a && b && c, "null device pointer");
    return ASMC_OK;
}

// ---------------------------------------------------------------------------------------------
// Device-side k-ary bisection (smc/base.py:167-186): the whole adaptive-beta search runs as a chain of
// launches without host round trips.  State (doubles) in ctx->d_small + 2048:
//   [0] beta_min  [1] beta_max  [2] done  [3] target_eff  [4] tol  [5] log N  [6] n_pass  [7] beta0
//   [8] N  [9] ESS(1)/N  [16..27] bracket and next grid as dyadic node indices (asmc_bisect.h)
#define BIS_LEVELS 4
#define BIS_NODES 15

__device__ __forceinline__ double ess_over_n(double m, double S1, double S2, double logN, double N) {
    // the host's smc_math.ess (utils.py:510-512 on samples.py:1244-1249), same operation order
    const double c = (m + log(S1)) - logN;
    const double mp = m + c;
    const double l1 = mp + log(S1);
    const double l2 = mp * 2.0 + log(S2);
    return exp(l1 * 2.0 - l2) / N;
}

struct BisInit {
    double beta0, target, tol, logN, N;
    double plain;  // 1: no prediction windows (ASMC_BISECT_PLAIN: the 16-ary search of rounds 1-2, for comparison)
};
static double bis_plain_mode() {
    static const double v = getenv("ASMC_BISECT_PLAIN") ? 1.0 : 0.0;
    return v;
}

// Closes a search round: fixed-order reduction of the block partials, ESS of every candidate (one lane each), the
// decisions and the sixteen nodes of the next round (asmc_bisect.h: plain 16-ary step or a window around the predicted
// root).  Runs in the LAST block of the round's reduction kernel (k_bis_sums).
//   first round: the 15 heap-ordered midpoints of [beta0, 1] AND beta = 1 itself (16th column; smc/base.py:170-175:
//   eff(1) >= target ends the search at beta* = 1); this round also creates the state record.
//   Stabilising maxima are not searched for: at beta >= beta0 every log-weight is (beta - beta0) * Delta_i up to
//   rounding, so max_i lw_i(beta) = m(1) (beta - beta0)/(1 - beta0) to rounding as well (m(1): exact, from the max
//   kernel), and a log-sum-exp only needs a shift near the maximum, not the maximum itself.
// State st[]: [0] beta_min [1] beta_max [2] done [3] target_eff [4] tol [5] log N [6] rounds [7] beta0 [8] N
//   [9] ESS(1)/N [10] m(1) [11..13] (m, S1, S2) at beta_min [14] 1 when [11..13] are valid [15] NaN log-weights
//   (sharded search only) [16..27] the planner's bracket / grid in node indices (asmc_bisect.h)
//   [32] S1(1) [33] S2(1)  [34..38] next round's grid: c1, c2, m of the LOWEST candidate, spacing h, Delta_max
// Fixed-order reduction of the block partial records (32 columns) into s_S; valid after the next __syncthreads().
__device__ __forceinline__ void bis_reduce_partials(const double* __restrict__ partials, int nblocks,
                                                    double (*s_red)[33], double* s_S) {
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5, nparts = blockDim.x >> 5;
    double v = 0.0;
    for (int b0 = part; b0 < nblocks; b0 += 16 * nparts) {  // sixteen records in flight per thread, fixed order
        double t16[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int b = b0 + q * nparts;
            t16[q] = b < nblocks ? partials[(size_t)b * 32 + col] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) v += t16[q];
    }
    s_red[part][col] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int q = 0; q < nparts; q++) t += s_red[q][threadIdx.x];
        s_S[threadIdx.x] = t;
    }
}

// The same reduction for records whose sums were shifted by each block's OWN maximum (first round of k_is_weights: no
// grid-wide maximum pass in front of it).  The first round's candidates are beta0 + ks (1 - beta0) / 16, ks = 1 .. 16, so
// record b, candidate ks, power pw in {1, 2} is rescaled to the merged maximum M by
//     exp(lw - m_b t) = exp(lw - M t) exp((m_b - M) / 16)^(pw ks),   t = ks / 16
// (what k_bis_decide does with the ranks' records of a sharded search): ONE exponential per record and an integer power.
// `mrec`: (m_b, NaN count) per block; a block without a finite log-weight (its sums are zero) counts as m_b = M.
// Also returns the merged maximum and the NaN census in s_mn[0], s_mn[1].  nblocks <= 512.
__device__ __forceinline__ void bis_reduce_partials_scaled(const double* __restrict__ partials, const double* __restrict__ mrec,
                                                           int nblocks, double (*s_red)[33], double* s_S, double* s_base,
                                                           double* s_mn) {
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5, nparts = blockDim.x >> 5;
    // sorted position (1 .. 16) of heap column k: the inverse of bis_col_of_sorted
    const int k = col >> 1;
    int ks = 16;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (bis_col_of_sorted(j) == k) ks = j + 1;
    double t16[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int b = part + q * nparts;
        t16[q] = b < nblocks ? partials[(size_t)b * 32 + col] : 0.0;
    }
    const int b_own = threadIdx.x;  // blockDim.x = 512 >= nblocks
    const double m_own = b_own < nblocks ? mrec[2 * b_own] : -INFINITY, n_own = b_own < nblocks ? mrec[2 * b_own + 1] : 0.0;
    {
        double v = wave_max(m_own == m_own ? m_own : -INFINITY), c = wave_sum(n_own);  // (integer-valued counts: any order)
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][0] = v, s_red[threadIdx.x >> 6][1] = c;
    }
    __syncthreads();
    double m_all = -INFINITY, nan_total = 0.0;
    for (int x = 0; x < (int)(blockDim.x >> 6); x++) m_all = fmax(m_all, s_red[x][0]), nan_total += s_red[x][1];
    if (threadIdx.x == 0) s_mn[0] = m_all, s_mn[1] = nan_total;
    s_base[b_own] = (m_own > -INFINITY && m_all > -INFINITY) ? exp((m_own - m_all) * (1.0 / 16.0)) : 1.0;
    __syncthreads();
    const int e0 = ((col & 1) ? 2 : 1) * ks;  // 1 .. 32
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int b = part + q * nparts;
        double f = 1.0, pw = b < nblocks ? s_base[b] : 1.0;
#pragma unroll
        for (int bit = 0; bit < 6; bit++) {
            if ((e0 >> bit) & 1) f *= pw;
            pw *= pw;
        }
        v += t16[q] * f;
    }
    s_red[part][col] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        double tt = 0.0;
        for (int q = 0; q < nparts; q++) tt += s_red[q][threadIdx.x];
        s_S[threadIdx.x] = tt;
    }
}

// Block-local copy of the search state: the record st[0..39] (layout above) and the candidates' c1, c2, m, shift
// (in: m of this round's candidates; out: those of the next round)
struct BisLds {
    double st[40];
    double bp[4][16];
};

__device__ __forceinline__ void bis_lds_init(BisLds& L, const BisInit& init, double m_one, double nan_total) {
    if (threadIdx.x < 40) {
        const int i = threadIdx.x;
        L.st[i] = i == 0 ? init.beta0 : i == 1 ? 1.0 : i == 3 ? init.target : i == 4 ? init.tol : i == 5 ? init.logN
                  : i == 7 ? init.beta0 : i == 8 ? init.N : i == 10 ? m_one : i == 15 ? nan_total : i == BIS_MODE ? init.plain : 0.0;
    }
}

// Closes a round on the block-local state.  Precondition: L and s_S are filled and a __syncthreads() lies behind
// that; returns behind a __syncthreads() with L.st[2] (done), L.st[39] (a new candidate grid was chosen) up to date.
// Lanes 0-15: the floats of this round's sixteen nodes (asmc_bisect.h: bis_node_beta - exactly the values the sequential
// loop would hold), their shifts and ESS/N; lane 0: the decisions and the next grid (bis_plan).
__device__ __forceinline__ void bis_tail_core(BisLds& L, bool first, const double* s_S, double* s_eff) {
    double(&s_st)[40] = L.st;
    if (threadIdx.x < 16) {
        const int j = threadIdx.x;
        const double beta0 = s_st[BIS_BETA0], m_one = s_st[BIS_M_ONE];
        const int LU = first ? 4 : (int)s_st[BIS_LU];
        const long long K = first ? (long long)(j + 1) : (long long)s_st[BIS_KFIRST] + j * (long long)s_st[BIS_STRIDE];
        const double b = bis_node_beta(K, LU, beta0);
        const double mj = m_one * ((b - beta0) * (1.0 / (1.0 - beta0)));
        const int c = bis_col_of_sorted(j);
        const double e = ess_over_n(mj, s_S[2 * c], s_S[2 * c + 1], s_st[BIS_LOGN], s_st[BIS_N]);
        L.bp[0][j] = b;
        L.bp[1][j] = log(e) - log(s_st[BIS_TARGET]);
        L.bp[2][j] = mj;
        s_eff[j] = e;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[40];  // the record in registers: the decision chain pays no LDS round trip per access
#pragma unroll
        for (int i = 0; i < 40; i++) r[i] = s_st[i];
        bis_plan(r, first, L.bp[0], L.bp[2], s_eff, L.bp[1], s_S);
#pragma unroll
        for (int i = 0; i < 40; i++) s_st[i] = r[i];
    }
    __syncthreads();
}

// Round tail on the state record in global memory (one launch per round: k_bis_sums' last block, k_bis_decide).
// `partials` == nullptr: s_S already holds the column sums (sharded search: the rank records merged by k_bis_decide).
__device__ __forceinline__ void bis_tail_body(double* __restrict__ st, const double* __restrict__ partials, int nblocks,
                                              bool first, double m_one_in, const BisInit& init, double (*s_red)[33],
                                              double* s_S, double* s_eff, double nan_total = 0.0) {
    // the state record is staged in LDS: lane 0's decision chain must not pay a global-memory latency per dependent access
    __shared__ BisLds L;
    if (first)
        bis_lds_init(L, init, m_one_in, nan_total);
    else if (threadIdx.x < 40)
        L.st[threadIdx.x] = st[threadIdx.x];
    if (partials) bis_reduce_partials(partials, nblocks, s_red, s_S);
    __syncthreads();
    bis_tail_core(L, first, s_S, s_eff);
    if (threadIdx.x < 40) st[threadIdx.x] = L.st[threadIdx.x];
}

// heap node k of the bisection tree rooted at (lo, hi): the midpoint the sequential loop would try there
__device__ __forceinline__ double bis_heap_mid(int k, double lo, double hi) {
    const int kp = k + 1;
    const int depth = 31 - __clz(kp);
    double mid = 0.5 * (hi + lo);
    for (int lev = depth - 1; lev >= 0; lev--) {
        if ((kp >> lev) & 1)
            lo = mid;  // right child: eff >= target there
        else
            hi = mid;
        mid = 0.5 * (hi + lo);
    }
    return mid;
}

// Sharded search, the decide half of round `round` on a BLOCK-LOCAL state (what k_bis_decide does on the record in global
// memory): merges the ranks' records in rank order and closes the round.  Every block of the NEXT kernel of the chain runs
// it redundantly from the same gathered records (fixed order: the same bits everywhere), so the round costs no launch of its
// own.  `st_prev`: the state record after the previous round (unused in round 0, which creates it).  Returns behind a
// __syncthreads(); a search that had converged earlier passes through unchanged.
__device__ __forceinline__ void bis_shard_decide(BisLds& L, const double* __restrict__ recs, int world, int round,
                                                 const BisInit& init, const double* __restrict__ st_prev, double* s_S,
                                                 double* s_eff) {
    if (round > 0) {
        if (threadIdx.x < 40) L.st[threadIdx.x] = st_prev[threadIdx.x];
        __syncthreads();
        if (L.st[2] != 0.0) return;  // converged earlier (uniform: every thread reads the same value)
    }
    double m_all = recs[32], nan_total = 0.0;
    for (int r = 1; r < world; r++) m_all = fmax(m_all, recs[(size_t)r * ASMC_BIS_REC + 32]);
    if (round == 0)
        for (int r = 0; r < world; r++) nan_total += recs[(size_t)r * ASMC_BIS_REC + 33];
    if (threadIdx.x < 32) {
        const int c = threadIdx.x, k = c >> 1;
        double acc = 0.0;
        if (round == 0) {
            const double beta = k < BIS_NODES ? bis_heap_mid(k, init.beta0, 1.0) : 1.0;
            const double t = (beta - init.beta0) * (1.0 / (1.0 - init.beta0));
            const double pw = (c & 1) ? 2.0 : 1.0;
            for (int r = 0; r < world; r++) {
                const double mr = recs[(size_t)r * ASMC_BIS_REC + 32];
                acc += recs[(size_t)r * ASMC_BIS_REC + c] * exp(pw * (mr * t - m_all * t));
            }
        } else {
            for (int r = 0; r < world; r++) acc += recs[(size_t)r * ASMC_BIS_REC + c];
        }
        s_S[c] = acc;
    }
    if (round == 0) bis_lds_init(L, init, m_all, nan_total);
    __syncthreads();
    bis_tail_core(L, round == 0, s_S, s_eff);
}

// One bisection round in ONE launch.  The 15 candidates and the bracket's upper end are equally spaced,
// beta_j = beta_1 + (j-1) h (j = 1..16), and every log-sum-exp is shifted by m_j = (beta_j - beta0) Delta_max, so for
// particle i
//     exp(lw_i(beta_j) - m_j) = exp(lw_i(beta_1) - m_1) * r_i^(j-1),   r_i = exp(h (Delta_i - Delta_max)) <= 1:
// two exponentials per particle instead of sixteen.  The first factor uses the reference's own expression for the
// log-weight (samples.py:1222-1224); the progression is non-increasing in j, so it can neither overflow nor lose
// a term that matters.  Relative deviation from direct exponentials: ~1e-14 (it decides `eff >= target`
// comparisons only; the values reported at the chosen beta carry the same 1e-14).
// The block that arrives last (ticket counter behind an agent-scope release) reduces the partials and runs the
// tail, so a round costs one launch; the first round derives its grid from m(1) (max kernel) itself.
#define BIS_THREADS 512  // one block per CU: few, large partial records keep the last block's reduction short

__global__ __launch_bounds__(BIS_THREADS) void k_bis_sums(int64_t n, const double* __restrict__ ll,
                                                        const double* __restrict__ lp, const double* __restrict__ lq,
                                                        double* __restrict__ st,
                                                        double* partials, unsigned int* ticket, int round, BisInit init,
                                                        const unsigned long long* __restrict__ keys,
                                                        double* __restrict__ rec_out,
                                                        const double* __restrict__ recs_prev = nullptr, int world = 0,
                                                        const double* __restrict__ st_prev = nullptr) {
    __shared__ double s_red[BIS_THREADS / 32][33];
    __shared__ double s_S[32];
    __shared__ double s_eff[16];
    __shared__ int s_last;
    __shared__ BisLds Lf;  // folded sharded rounds (recs_prev given): the state after the previous round's decide half
    double c1, c2, m1, h, dmax, m_one = 0.0, m_base = 0.0;
    if (recs_prev) {
        // round >= 1 of the one-chain sharded step: the previous round is closed HERE, by every block, from the gathered
        // records; block 0 leaves the state in `st` (the other of the two state buffers: nobody reads what is being written)
        bis_shard_decide(Lf, recs_prev, world, round - 1, init, st_prev, s_S, s_eff);
        if (blockIdx.x == 0 && threadIdx.x < 40) st[threadIdx.x] = Lf.st[threadIdx.x];
        if (Lf.st[2] != 0.0) return;
        c1 = Lf.st[34], c2 = Lf.st[35], m1 = Lf.st[36], h = Lf.st[37], dmax = Lf.st[38], m_base = Lf.st[10];
        __syncthreads();  // (s_S, s_eff are reused by the last block below)
    } else if (round == 0) {
        m_one = key_to_f64(keys[0]);
        const double lo = init.beta0;
        double b1 = 1.0;
        for (int lev = 0; lev < BIS_LEVELS; lev++) b1 = 0.5 * (b1 + lo);  // leftmost leaf of the first tree
        const double inv = 1.0 / (1.0 - lo);
        c1 = lo - b1, c2 = b1 - lo, m1 = m_one * ((b1 - lo) * inv), h = (1.0 - lo) / 16.0, dmax = m_one * inv;
    } else {
        if (st[2] != 0.0) return;  // converged in an earlier round (uniform across the grid)
        c1 = st[34], c2 = st[35], m1 = st[36], h = st[37], dmax = st[38];
    }
    double s1[16], s2[16];
#pragma unroll
    for (int j = 0; j < 16; j++) s1[j] = 0.0, s2[j] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * BIS_THREADS;
    // two particles per trip: six loads in flight before the first exponential
    for (int64_t i = (int64_t)blockIdx.x * BIS_THREADS + threadIdx.x; i < n; i += 2 * stride) {
        const int64_t i2 = i + stride;
        const bool has2 = i2 < n;
        const double a = ll[i], b = lp[i], q = lq[i];
        const double a2 = has2 ? ll[i2] : 0.0, b2 = has2 ? lp[i2] : 0.0, q2 = has2 ? lq[i2] : 0.0;
        double e = exp(lw_of(a, b, q, c1, c2) - m1);
        const double r = exp(h * (((a + b) - q) - dmax));
        double e2 = has2 ? exp(lw_of(a2, b2, q2, c1, c2) - m1) : 0.0;
        const double r2 = has2 ? exp(h * (((a2 + b2) - q2) - dmax)) : 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            s1[j] += e;
            s2[j] += e * e;
            e *= r;
            s1[j] += e2;
            s2[j] += e2 * e2;
            e2 *= r2;
        }
    }
    __shared__ double s_p[BIS_THREADS / 64][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave-level reduction of the 32 accumulators as a butterfly reduce-scatter: at distance o a lane keeps one
    // half of its values and trades the other half with its partner, so 16+8+4+2+1+1 = 32 values cross lanes
    // instead of 32 x 6 (the cross-lane permutes run on the LDS pipe, which 8 waves per CU share; the plain
    // per-value butterflies cost more than the two exponentials per particle).  vals[c] = column c of the
    // partial record (2 * heap index + {0: S1, 1: S2}; heap index 15 = the bracket's upper end); lane l ends up with
    // the wave total of column l >> 1.
    constexpr int HEAP_OF_SORTED[16] = {7, 3, 8, 1, 9, 4, 10, 0, 11, 5, 12, 2, 13, 6, 14, 15};  // in-order -> heap index
    double vals[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        vals[2 * HEAP_OF_SORTED[j]] = s1[j];
        vals[2 * HEAP_OF_SORTED[j] + 1] = s2[j];
    }
#pragma unroll
    for (int o = 32, cnt = 32; o >= 2; o >>= 1, cnt >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int k = 0; k < cnt / 2; k++) {
            const double keep = up ? vals[k + cnt / 2] : vals[k];
            const double send = up ? vals[k] : vals[k + cnt / 2];
            vals[k] = keep + __shfl_xor(send, o, 64);
        }
    }
    vals[0] += __shfl_xor(vals[0], 1, 64);
    if ((lane & 1) == 0) s_p[wave][lane >> 1] = vals[0];
    __syncthreads();
    if (threadIdx.x < 32) {
        double v = s_p[0][threadIdx.x];
        for (int w = 1; w < BIS_THREADS / 64; w++) v += s_p[w][threadIdx.x];
        partials[(size_t)blockIdx.x * 32 + threadIdx.x] = v;
    }
    // In-launch hand-off to the last-arriving block (cdna_hip_programming.md Guideline 16, counter form): the storing
    // wave drains its stores, the block meets, ONE lane releases at agent scope and draws a ticket.  The counter is
    // zeroed once per search by a memset on the stream and only ever grows: round r ends at ticket (r+1) * grid - 1.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (unsigned int)(round + 1) * gridDim.x - 1u);
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    if (rec_out) {
        // sharded search: this rank's column sums, the shift base they are relative to (local m(1) in the first
        // round, the merged one afterwards) and the local NaN census leave as one record; k_bis_decide closes the round
        // on every rank after the all-gather
        bis_reduce_partials(partials, (int)gridDim.x, s_red, s_S);
        __syncthreads();
        if (threadIdx.x < 32) rec_out[threadIdx.x] = s_S[threadIdx.x];
        if (threadIdx.x == 32) rec_out[32] = round == 0 ? m_one : recs_prev ? m_base : st[10];
        if (threadIdx.x == 33) rec_out[33] = round == 0 ? (double)keys[ASMC_MAX_BETAS] : 0.0;
        return;
    }
    bis_tail_body(st, partials, (int)gridDim.x, round == 0, m_one, init, s_red, s_S, s_eff);
}

// ---------------------------------------------------------------------------------------------------------------
// The importance step's weight half in ONE persistent launch (asmc_importance_weights; smc/base.py:167-186 followed by
// samples.py:1230-1249 and :1276-1277): exact maximum at beta = 1 and the NaN census, the k-ary search rounds of
// k_bis_sums, the evidence-variance / second log-sum-exp pass of k_weights_m2_lse and the normalised weights with the
// scan's tile sums.  One block per CU, every block resident (the launch asks for no more blocks than CUs), the phases
// separated by grid barriers on a counter that only grows (no memset node; the launch's budget of ISW_BARRIERS
// arrivals per block is topped up at the end so that the next launch's base is where the counter stands).  A block
// owns a contiguous range of whole 4096-particle chunks (two scan tiles); REG: one chunk per block, the particles'
// (ll + lp, lq) stay in registers across all phases, so HBM/L2 is read once instead of once per round.
// Every block closes a round redundantly from the same partial records (fixed order => the same bits everywhere):
// no block waits for another one's decision, and the state record never leaves LDS between rounds.
// Results: st_out[0..39] = the search record (layout above), [40] sum (exp(lw - m) - mean_u)^2, [41] S1', [42] lse',
// [43] shift, [44] mp, [45] 1 when beta* was found and w / tile sums are those of beta* (otherwise w = 1/N).
#define ISW_THREADS 512
#define ISW_PER 8
#define ISW_CHUNK (ISW_THREADS * ISW_PER)
#define ISW_MAX_ROUNDS 20
#define ISW_BARRIERS 32

// Two-level arrival: block b counts in on the counter of its group b % ISW_GROUPS (the counters sit in different
// memory channels, so the same-address atomics of one arrival wave serialise 8-fold less); the last block of a group
// counts the group in on the top counter, which everybody polls with plain device-scope loads.
// bar: [0] top counter, [ISW_BAR_STRIDE * (1 + g)] group g; bases: their values when this launch started.
#define ISW_GROUPS 8
#define ISW_BAR_STRIDE 1024  // counters 4 KB apart
struct IswBases {
    unsigned int top, group[ISW_GROUPS];
};
// Returns false when the other blocks did not arrive within ~0.5 s: the launch was not fully resident (two such kernels
// of different processes sharing the GPU can each hold half of the CUs).  The caller then abandons the search - uniform
// weights, found = 0 - and raises the poison cell; the host resets the counters and stops using this kernel.
#define ISW_POISON_CELL (ISW_BAR_STRIDE * (ISW_GROUPS + 2))
#define ISW_SPIN_LIMIT (1 << 18)
__device__ __forceinline__ bool isw_barrier(unsigned int* bar, const IswBases& bases, unsigned int k, int G) {
    __shared__ int s_ok;
    // only wave 0 writes what the other blocks read behind the barrier (partial records, block maxima): the other waves'
    // stores (the gather's records, the weights) need not have landed
    if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x % ISW_GROUPS;
        const unsigned int ngroups = G < ISW_GROUPS ? G : ISW_GROUPS;
        const unsigned int gsize = (unsigned int)(G - g + ISW_GROUPS - 1) / ISW_GROUPS;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned int old = __hip_atomic_fetch_add(bar + ISW_BAR_STRIDE * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old - bases.group[g] == k * gsize - 1u)  // this block completes its group's k-th arrival
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = bases.top + k * ngroups;
        int spins = 0;
        while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && spins < ISW_SPIN_LIMIT) {
            __builtin_amdgcn_s_sleep(1);
            spins++;
        }
        s_ok = spins < ISW_SPIN_LIMIT;
        if (!s_ok) __hip_atomic_store(bar + ISW_POISON_CELL, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return s_ok != 0;
}

template <bool REG>
__global__ __launch_bounds__(ISW_THREADS) void k_is_weights(int64_t n, const double* __restrict__ ll,
                                                           const double* __restrict__ lp, const double* __restrict__ lq,
                                                           double* __restrict__ st_out, double* partials,
                                                           unsigned int* bar, IswBases bar_bases, BisInit init,
                                                           int64_t chunk, double* __restrict__ w,
                                                           double* __restrict__ tiles, double* __restrict__ rec) {
    __shared__ BisLds L;
    __shared__ double s_red[ISW_THREADS / 32][33];
    __shared__ double s_S[32];
    __shared__ double s_eff[16];
    __shared__ double s_p[ISW_THREADS / 64][32];
    __shared__ double s_sc[8];
    __shared__ double s_base[ISW_THREADS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = (int)gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    const int nsub = (int)(chunk / ISW_CHUNK);
    unsigned int nbar = 0;
#ifdef ISW_STAMP
    const unsigned long long t_start = wall_clock64();
    int n_stamp = 0;
    double stamps[16];
#define ISW_MARK()                                                                         \
    do {                                                                                   \
        if (n_stamp < 16) stamps[n_stamp++] = (double)(wall_clock64() - t_start) * 0.01; \
    } while (0)
#else
#define ISW_MARK() \
    do {           \
    } while (0)
#endif
    bool poisoned = false;  // a barrier timed out: finish with uniform weights (see isw_barrier)
    auto barrier = [&]() {
        if (poisoned) return;
        nbar++;
        poisoned = !isw_barrier(bar, bar_bases, nbar, G);
    };
    auto pbuf = [&](unsigned int k) { return partials + (size_t)(k & 1u) * (size_t)G * 32; };
    double* mrec = partials + (size_t)2 * G * 32;  // (maximum, NaN count) of every block

    // REG: the block's particles as (ll + lp, lq) in LDS, staged once (64 KB of dynamic LDS)
    extern __shared__ double s_part[];
    double* s_ab = s_part;
    double* s_q = s_part + ISW_CHUNK;
    if (REG) {
#pragma unroll
        for (int k = 0; k < ISW_PER; k++) {
            const int64_t i = lo + k * ISW_THREADS + tid;
            const bool ok = i < hi;
            const double a = ok ? ll[i] : 0.0, b = ok ? lp[i] : 0.0, q = ok ? lq[i] : 0.0;
            s_ab[k * ISW_THREADS + tid] = a + b;
            s_q[k * ISW_THREADS + tid] = q;
            // the (ll, lp, lq, 0) record asmc_gather reads per draw (k_pack_records' job, done on the way)
            if (rec && ok) *reinterpret_cast<double4*>(rec + 4 * i) = make_double4(a, b, q, 0.0);
        }
        // every thread only ever reads back its own entries: no barrier needed
    }
    // particle k of sub-chunk `sub` of this thread: (ll + lp, lq); false behind the end of the block's range
    auto get = [&](int sub, int k, double& abv, double& qv) {
        const int64_t i = lo + (int64_t)sub * ISW_CHUNK + k * ISW_THREADS + tid;
        const bool ok = i < hi;
        if (REG) {
            abv = s_ab[k * ISW_THREADS + tid];
            qv = s_q[k * ISW_THREADS + tid];
        } else {
            const double a = ok ? ll[i] : 0.0, b = ok ? lp[i] : 0.0;
            abv = a + b;
            qv = ok ? lq[i] : 0.0;
        }
        return ok;
    };
    // the reference's log-weight (samples.py:1222-1224) from ll + lp and lq
    auto lw_at = [&](double abv, double qv, double c1, double c2) {
        const double t1 = c1 * qv;
        const double t2 = c2 * abv;
        return t1 + t2;
    };

    // ---- phase 0: THIS BLOCK's maximum of the log-weights at beta = 1 and its NaN census (no grid barrier: the first
    // search round shifts by the block's own maximum and the round's reduction rescales to the merged one) -------------
    {
        const double c1 = init.beta0 - 1.0, c2 = 1.0 - init.beta0;
        double mx = -INFINITY;
        long long nn = 0;
        for (int sub = 0; sub < nsub; sub++) {
#pragma unroll 4
            for (int k = 0; k < ISW_PER; k++) {
                double abv, qv;
                bool ok;
                if (REG) {
                    ok = get(sub, k, abv, qv);
                } else {  // streaming: the records are packed from this first read
                    const int64_t i = lo + (int64_t)sub * ISW_CHUNK + k * ISW_THREADS + tid;
                    ok = i < hi;
                    const double a = ok ? ll[i] : 0.0, b = ok ? lp[i] : 0.0;
                    abv = a + b, qv = ok ? lq[i] : 0.0;
                    if (rec && ok) *reinterpret_cast<double4*>(rec + 4 * i) = make_double4(a, b, qv, 0.0);
                }
                if (ok) {
                    const double lw = lw_at(abv, qv, c1, c2);
                    if (lw != lw)
                        nn++;
                    else
                        mx = fmax(mx, lw);
                }
            }
        }
        mx = wave_max(mx);
        nn = wave_sum_ll(nn);
        if (lane == 0) s_p[wave][0] = mx, s_p[wave][1] = (double)nn;
        __syncthreads();
        if (tid == 0) {
            double v = s_p[0][0], c = s_p[0][1];
            for (int x = 1; x < ISW_THREADS / 64; x++) v = fmax(v, s_p[x][0]), c += s_p[x][1];
            s_sc[0] = v, s_sc[1] = c;
            mrec[2 * blockIdx.x] = v, mrec[2 * blockIdx.x + 1] = c;
        }
        __syncthreads();
        ISW_MARK();
    }
    const double m_block = s_sc[0] > -INFINITY ? s_sc[0] : 0.0;  // (no finite log-weight in the block: its sums are zero)
    double m_one = 0.0, nan_total = 0.0;
    bis_lds_init(L, init, 0.0, 0.0);  // (a poisoned first barrier leaves the loop before the record is created)

    // ---- phase 1: the search rounds (k_bis_sums on the resident particles) ------------------------------------------
    for (int round = 0; round < ISW_MAX_ROUNDS; round++) {
        double c1, c2, m1, h, dmax;
        if (round == 0) {
            const double lo0 = init.beta0;
            double b1 = 1.0;
            for (int lev = 0; lev < BIS_LEVELS; lev++) b1 = 0.5 * (b1 + lo0);  // leftmost leaf of the first tree
            const double inv = 1.0 / (1.0 - lo0);
            c1 = lo0 - b1, c2 = b1 - lo0, m1 = m_block * ((b1 - lo0) * inv), h = (1.0 - lo0) / 16.0, dmax = m_block * inv;
        } else {
            c1 = L.st[34], c2 = L.st[35], m1 = L.st[36], h = L.st[37], dmax = L.st[38];
        }
        double s1[16], s2[16];
#pragma unroll
        for (int j = 0; j < 16; j++) s1[j] = 0.0, s2[j] = 0.0;
        for (int sub = 0; sub < nsub; sub++) {
#pragma unroll 1
            for (int k = 0; k < ISW_PER; k += 2) {  // two particles per trip (k_bis_sums)
                double ab0, q0, ab1, q1;
                const bool v0 = get(sub, k, ab0, q0), v1 = get(sub, k + 1, ab1, q1);
                double e = v0 ? exp(lw_at(ab0, q0, c1, c2) - m1) : 0.0;
                const double r = v0 ? exp(h * ((ab0 - q0) - dmax)) : 0.0;
                double e2 = v1 ? exp(lw_at(ab1, q1, c1, c2) - m1) : 0.0;
                const double r2 = v1 ? exp(h * ((ab1 - q1) - dmax)) : 0.0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    s1[j] += e;
                    s2[j] += e * e;
                    e *= r;
                    s1[j] += e2;
                    s2[j] += e2 * e2;
                    e2 *= r2;
                }
            }
        }
        // wave-level butterfly reduce-scatter of the 32 accumulators (see k_bis_sums)
        constexpr int HEAP_OF_SORTED[16] = {7, 3, 8, 1, 9, 4, 10, 0, 11, 5, 12, 2, 13, 6, 14, 15};
        double vals[32];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            vals[2 * HEAP_OF_SORTED[j]] = s1[j];
            vals[2 * HEAP_OF_SORTED[j] + 1] = s2[j];
        }
#pragma unroll
        for (int o = 32, cnt = 32; o >= 2; o >>= 1, cnt >>= 1) {
            const bool up = (lane & o) != 0;
#pragma unroll
            for (int k = 0; k < cnt / 2; k++) {
                const double keep = up ? vals[k + cnt / 2] : vals[k];
                const double send = up ? vals[k] : vals[k + cnt / 2];
                vals[k] = keep + __shfl_xor(send, o, 64);
            }
        }
        vals[0] += __shfl_xor(vals[0], 1, 64);
        if ((lane & 1) == 0) s_p[wave][lane >> 1] = vals[0];
        __syncthreads();
        double* buf = pbuf(nbar + 1);
        if (tid < 32) {
            double v = s_p[0][tid];
            for (int x = 1; x < ISW_THREADS / 64; x++) v += s_p[x][tid];
            buf[(size_t)blockIdx.x * 32 + tid] = v;
        }
        ISW_MARK();
        barrier();
        if (poisoned) break;
        ISW_MARK();
        if (round == 0) {
            // merged maximum and NaN census of the blocks and the records rescaled to it; this creates the state record
            bis_reduce_partials_scaled(buf, mrec, G, s_red, s_S, s_base, s_sc);
            __syncthreads();
            m_one = s_sc[0], nan_total = s_sc[1];
            bis_lds_init(L, init, m_one, nan_total);
        } else {
            bis_reduce_partials(buf, G, s_red, s_S);
        }
        __syncthreads();
        bis_tail_core(L, round == 0, s_S, s_eff);
        ISW_MARK();
        if (L.st[2] != 0.0) break;  // uniform: every block took the same decisions
    }

    // ---- phase 2: evidence variance + second log-sum-exp at beta*, then the normalised weights -----------------------
    const double beta0 = init.beta0, beta = L.st[0], N = init.N;
    bool found = !poisoned && L.st[2] != 0.0 && L.st[14] != 0.0 && nan_total == 0.0 && beta > beta0;
    const double c1 = beta0 - beta, c2 = beta - beta0;
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    double shift = 0.0, mp = 0.0, lse = 0.0;
    if (found) {
        // smc_math.evidence_variance_and_lse's scalars, same operation order
        const double m = L.st[11], S1 = L.st[12];
        const double mean_u = S1 / N;
        shift = (m + log(S1)) - init.logN;
        mp = m + shift;
        double acc = 0.0, s1p = 0.0;
        for (int sub = 0; sub < nsub; sub++) {
#pragma unroll 2
            for (int k = 0; k < ISW_PER; k++) {
                double abv, qv;
                if (get(sub, k, abv, qv)) {
                    const double lw = lw_at(abv, qv, c1, c2);
                    const double dlt = exp(lw - m) - mean_u;
                    acc += dlt * dlt;
                    s1p += exp((lw + shift) - mp);
                }
            }
        }
        acc = wave_sum(acc);
        s1p = wave_sum(s1p);
        __syncthreads();
        if (lane == 0) s_p[wave][0] = acc, s_p[wave][1] = s1p;
        __syncthreads();
        double* buf = pbuf(nbar + 1);
        if (tid < 2) {
            double v = s_p[0][tid];
            for (int x = 1; x < ISW_THREADS / 64; x++) v += s_p[x][tid];
            buf[(size_t)blockIdx.x * 32 + tid] = v;
        }
        barrier();
        if (tid < 64) {  // k_finalize_columns' order: 64 strided chains, then the wave butterfly
            double v0 = 0.0, v1 = 0.0;
            for (int b = tid; b < G; b += 64) {
                v0 += buf[(size_t)b * 32];
                v1 += buf[(size_t)b * 32 + 1];
            }
            v0 = wave_sum(v0);
            v1 = wave_sum(v1);
            if (tid == 0) s_sc[2] = v0, s_sc[3] = v1, s_sc[4] = mp + log(v1);  // lse' = mp + log S1'
        }
        __syncthreads();
        lse = s_sc[4];
        if (poisoned) found = false;
    }
    const double w_uniform = 1.0 / N;
    for (int sub = 0; sub < nsub; sub++) {
        double ts0 = 0.0, ts1 = 0.0;
#pragma unroll 2
        for (int k = 0; k < ISW_PER; k++) {
            double abv, qv;
            if (get(sub, k, abv, qv)) {
                const double wv = found ? exp((lw_at(abv, qv, c1, c2) + shift) - lse) : w_uniform;
                w[lo + (int64_t)sub * ISW_CHUNK + k * ISW_THREADS + tid] = wv;
                if (k < ISW_PER / 2)
                    ts0 += wv;
                else
                    ts1 += wv;
            }
        }
        ts0 = wave_sum(ts0);
        ts1 = wave_sum(ts1);
        __syncthreads();
        if (lane == 0) s_p[wave][0] = ts0, s_p[wave][1] = ts1;
        __syncthreads();
        if (tid < 2) {
            double v = s_p[0][tid];
            for (int x = 1; x < ISW_THREADS / 64; x++) v += s_p[x][tid];
            const int64_t t = (lo + (int64_t)sub * ISW_CHUNK) / ASMC_SCAN_TILE + tid;
            if (t < n_tiles) tiles[t] = v;
        }
    }
    if (blockIdx.x == 0) {
        if (tid < 40) st_out[tid] = L.st[tid];
        if (tid == 40) st_out[40] = found ? s_sc[2] : 0.0;
        if (tid == 41) st_out[41] = found ? s_sc[3] : 0.0;
        if (tid == 42) st_out[42] = lse;
        if (tid == 43) st_out[43] = shift;
        if (tid == 44) st_out[44] = mp;
        if (tid == 45) st_out[45] = found ? 1.0 : 0.0;
#ifdef ISW_STAMP
        ISW_MARK();
        if (tid == 0)
            for (int k = 0; k < 16; k++) st_out[48 + k] = k < n_stamp ? stamps[k] : -1.0;
#endif
        // top the counters up to this launch's budget: the next launch's bases
        const unsigned int left = poisoned ? 0u : (unsigned int)(ISW_BARRIERS - nbar);  // (poisoned: the host resets them)
        if (tid == 0) __hip_atomic_fetch_add(bar, left * (unsigned int)(G < ISW_GROUPS ? G : ISW_GROUPS), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        if (tid >= 1 && tid <= ISW_GROUPS && tid - 1 < G) {
            const int g = tid - 1;
            __hip_atomic_fetch_add(bar + ISW_BAR_STRIDE * (1 + g), left * ((unsigned int)(G - g + ISW_GROUPS - 1) / ISW_GROUPS),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Sharded search, second half of a round (every rank runs it on the same all-gathered records, so every rank takes
// the same decisions): merges the G rank records in rank order and runs the tail of the single-rank kernel.  First
// round: the ranks reduced against their LOCAL m(1); the sums are rescaled to the merged maximum M = max_r m_r(1),
// exp(lw - m_r t) = exp(lw - M t) exp((M - m_r) t) with t = (beta - beta0)/(1 - beta0).  Later rounds use M on every rank.
__global__ __launch_bounds__(BIS_THREADS) void k_bis_decide(const double* __restrict__ recs, int world,
                                                          double* __restrict__ st, int round, BisInit init) {
    __shared__ double s_red[BIS_THREADS / 32][33];
    __shared__ double s_S[32];
    __shared__ double s_eff[16];
    if (round > 0 && st[2] != 0.0) return;  // converged earlier
    double m_all = recs[32], nan_total = 0.0;
    for (int r = 1; r < world; r++) m_all = fmax(m_all, recs[(size_t)r * ASMC_BIS_REC + 32]);
    if (round == 0)
        for (int r = 0; r < world; r++) nan_total += recs[(size_t)r * ASMC_BIS_REC + 33];
    if (threadIdx.x < 32) {
        const int c = threadIdx.x, k = c >> 1;
        double acc = 0.0;
        if (round == 0) {
            const double beta = k < BIS_NODES ? bis_heap_mid(k, init.beta0, 1.0) : 1.0;
            const double t = (beta - init.beta0) * (1.0 / (1.0 - init.beta0));
            const double pw = (c & 1) ? 2.0 : 1.0;
            for (int r = 0; r < world; r++) {
                const double mr = recs[(size_t)r * ASMC_BIS_REC + 32];
                acc += recs[(size_t)r * ASMC_BIS_REC + c] * exp(pw * (mr * t - m_all * t));
            }
        } else {
            for (int r = 0; r < world; r++) acc += recs[(size_t)r * ASMC_BIS_REC + c];
        }
        s_S[c] = acc;
    }
    __syncthreads();
    bis_tail_body(st, nullptr, 0, round == 0, m_all, init, s_red, s_S, s_eff, nan_total);
}

// Sharded importance step: k_weights_m2_lse and k_weights_map<1> with their scalars taken from the search state the last k_bis_decide left
// on the device (st[BIS_*], identical on every rank) - no host decision sits between the search, the evidence moments and
// the weights.  The scalars are formed in smc_math.resample_owner's operation order (= k_is_weights' phase 2).
struct ShardScalars {
    double c1, c2, m, mean_u, shift, mp;
    bool found;
};
__device__ __forceinline__ ShardScalars shard_scalars(const double* __restrict__ st) {
    ShardScalars s;
    const double beta0 = st[BIS_BETA0], beta = st[BIS_BMIN], N = st[BIS_N];
    s.found = st[BIS_DONE] != 0.0 && st[BIS_TRIP_OK] != 0.0 && st[BIS_NAN] == 0.0 && beta > beta0;
    s.c1 = beta0 - beta, s.c2 = beta - beta0;
    s.m = st[BIS_TRIP_M];
    const double S1 = st[BIS_TRIP_S1];
    s.mean_u = S1 / N;
    s.shift = (s.m + log(S1)) - st[BIS_LOGN];
    s.mp = s.m + s.shift;
    return s;
}

__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_m2_lse_shard(int64_t n, const double* __restrict__ ll,
                                                                    const double* __restrict__ lp,
                                                                    const double* __restrict__ lq,
                                                                    const double* __restrict__ st,
                                                                    double* partials, unsigned int* ticket,
                                                                    double* __restrict__ out,
                                                                    const double* __restrict__ recs_last, int world,
                                                                    int last_round, BisInit init, double* __restrict__ st_out) {
    __shared__ BisLds Lf;
    __shared__ double s_S[32];
    __shared__ double s_eff[16];
    if (recs_last) {  // the search's last round is closed here (bis_shard_decide); block 0 leaves the final state in st_out
        bis_shard_decide(Lf, recs_last, world, last_round, init, st, s_S, s_eff);
        if (blockIdx.x == 0 && threadIdx.x < 40) st_out[threadIdx.x] = Lf.st[threadIdx.x];
    }
    const ShardScalars s = shard_scalars(recs_last ? Lf.st : st);
    double acc = 0.0, s1p = 0.0;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    if (s.found)
        for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
            const double lw = lw_of(ll[i], lp[i], lq[i], s.c1, s.c2);
            const double dlt = exp(lw - s.m) - s.mean_u;
            acc += dlt * dlt;
            s1p += exp((lw + s.shift) - s.mp);
        }
    __shared__ double s_p[ASMC_BLOCK / 64][2];
    acc = wave_sum(acc);
    s1p = wave_sum(s1p);
    if ((threadIdx.x & 63) == 0) {
        s_p[threadIdx.x >> 6][0] = acc;
        s_p[threadIdx.x >> 6][1] = s1p;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = s_p[0][threadIdx.x];
        for (int w = 1; w < ASMC_BLOCK / 64; w++) v += s_p[w][threadIdx.x];
        partials[(size_t)blockIdx.x * 2 + threadIdx.x] = v;
    }
    // k_finalize_columns by the block that arrives last (k_bis_sums' hand-off; the counter resets itself for the next call)
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1u);
        if (s_last) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 128) return;
    {
        const int col = threadIdx.x >> 6, lane = threadIdx.x & 63;
        double v = 0.0;
        for (int b = lane; b < (int)gridDim.x; b += 64) v += __builtin_nontemporal_load(partials + (size_t)b * 2 + col);
        v = wave_sum(v);
        if (lane == 0) out[col] = v;
    }
}

// w = exp((lw + shift) - lse) with lse = mp + log(S1'), S1' = the ranks' second sums added in rank order (parts[world][2], the
// all-gathered outputs of the pass above); carry_out[0] = this rank's approximate incoming cdf sum (the lower ranks' share).
// A search that has not converged (or NaN weights, or an empty sum) leaves uniform weights and the uniform carry: everything
// enqueued behind stays well defined, the host discards it when it reads the state.
__global__ __launch_bounds__(ASMC_BLOCK) void k_weights_map_shard(int64_t n, const double* __restrict__ ll,
                                                                 const double* __restrict__ lp,
                                                                 const double* __restrict__ lq,
                                                                 const double* __restrict__ st,
                                                                 const double* __restrict__ parts, int world, int rank,
                                                                 double carry_uniform, double* __restrict__ out,
                                                                 double* __restrict__ carry_out,
                                                                 double* __restrict__ tile_sums,
                                                                 double* __restrict__ st_copy, double* __restrict__ rec) {
    const ShardScalars s = shard_scalars(st);
    double s1p = parts[1], below = 0.0;
    for (int r = 1; r < world; r++) {
        if (r == rank) below = s1p;
        s1p += parts[2 * r + 1];
    }
    const bool found = s.found && s1p > 0.0 && s1p < INFINITY;
    const double lse = s.mp + log(s1p);
    const double w_uniform = 1.0 / st[BIS_N];
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) carry_out[0] = found ? below / s1p : carry_uniform;
        if (st_copy && threadIdx.x < 40) st_copy[threadIdx.x] = st[threadIdx.x];  // the search state, next to parts / info
    }
    // one block per scan tile (ASMC_SCAN_TILE particles): the tile's weight sum - the cdf's approximate prefix hints, any
    // order will do - leaves with the weights
    const int64_t base = (int64_t)blockIdx.x * ASMC_SCAN_TILE;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < ASMC_SCAN_TILE / ASMC_BLOCK; k++) {
        const int64_t i = base + k * ASMC_BLOCK + threadIdx.x;
        if (i < n) {
            const double a = ll[i], b = lp[i], q = lq[i];
            // the (ll, lp, lq, 0) record asmc_gather reads per draw (k_pack_records' job, done on the way: asmc_rec_claim)
            if (rec) *reinterpret_cast<double4*>(rec + 4 * i) = make_double4(a, b, q, 0.0);
            const double lw = lw_of(a, b, q, s.c1, s.c2) + s.shift;
            const double wv = found ? exp(lw - lse) : w_uniform;
            out[i] = wv;
            acc += wv;
        }
    }
    __shared__ double s_t[ASMC_BLOCK / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = s_t[0];
        for (int w = 1; w < ASMC_BLOCK / 64; w++) v += s_t[w];
        tile_sums[blockIdx.x] = v;
    }
}

extern "C" {

int asmc_weights_max(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                     double beta0, const double* betas_host, int K, double* m_host,
                     int64_t* n_nan_host, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(K >= 1 && K <= ASMC_MAX_BETAS && betas_host && m_host, "bad K / null host pointer");
    hipStream_t st = as_stream(stream);
    rc = launch_max(ctx, n, ll, lp, lq, beta0, betas_host, K, st);
    if (rc) return rc;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(ctx->h_pinned);
    ASMC_HIP(hipMemcpyAsync(h, ctx->d_keys, sizeof(unsigned long long) * (ASMC_MAX_BETAS + 1),
                            hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < K; k++) m_host[k] = key_to_f64(h[k]);
    if (n_nan_host) *n_nan_host = (int64_t)h[ASMC_MAX_BETAS];
    return ASMC_OK;
}

int asmc_weights_sums(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                      double beta0, const double* betas_host, const double* m_host,
                      const double* shift_host, int K, double* out_host, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(K >= 1 && K <= ASMC_MAX_BETAS && betas_host && m_host && out_host, "bad K / null host pointer");
    hipStream_t st = as_stream(stream);
    int grid = 0;
    rc = launch_sums(ctx, n, ll, lp, lq, beta0, betas_host, m_host, shift_host, K, false, &grid, st);
    if (rc) return rc;
    const int ncols = bucket_of(K) * 2;
    ASMC_LAUNCH(ctx, st, "k_finalize_columns", k_finalize_columns, dim3(ncols), dim3(64), 0, st, grid, ncols, ctx->d_partials,
                       ctx->d_small, 1, 0, (const unsigned long long*)nullptr,
                       (const unsigned long long*)nullptr);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(double) * ncols, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 2 * K; k++) out_host[k] = ctx->h_pinned[k];
    return ASMC_OK;
}

int asmc_weights_stats(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                       double beta0, const double* betas_host, int K, double* out_host,
                       asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(K >= 1 && K <= ASMC_MAX_BETAS && betas_host && out_host, "bad K / null host pointer");
    hipStream_t st = as_stream(stream);
    rc = launch_max(ctx, n, ll, lp, lq, beta0, betas_host, K, st);
    if (rc) return rc;
    int grid = 0;
    rc = launch_sums(ctx, n, ll, lp, lq, beta0, betas_host, nullptr, nullptr, K, true, &grid, st);
    if (rc) return rc;
    const int kt = bucket_of(K);
    ASMC_LAUNCH(ctx, st, "k_finalize_columns", k_finalize_columns, dim3(kt * 2), dim3(64), 0, st, grid, kt * 2, ctx->d_partials,
                       ctx->d_small, 1, 1, (const unsigned long long*)ctx->d_keys,
                       (const unsigned long long*)(ctx->d_keys + ASMC_MAX_BETAS));
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(double) * kt * 4, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 4 * K; k++) out_host[k] = ctx->h_pinned[k];
    return ASMC_OK;
}

int asmc_find_beta(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                   double target_eff, double tol, double* out_host, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(out_host != nullptr, "null host pointer");
    ASMC_REQUIRE(tol > 0.0 && beta0 >= 0.0 && beta0 < 1.0, "bad beta0 / tolerance");
    hipStream_t st = as_stream(stream);
    double* d_st = ctx->d_small + 2560;
    double* h = ctx->h_pinned + 4096 + 512;
    const BisInit init = {beta0, target_eff, tol, log((double)n), (double)n, bis_plain_mode()};
    // exact maximum at beta = 1 (also the NaN census: for beta > beta0 the NaN pattern of the log-weights does not
    // depend on beta); every round, the first one included (it evaluates beta = 1 as its 16th candidate), is then ONE launch
    const double one = 1.0;
    rc = launch_max(ctx, n, ll, lp, lq, beta0, &one, 1, st);
    if (rc) return rc;
    const int grid = grid_for(n, BIS_THREADS, ctx->num_cu);
    unsigned int* d_ticket = reinterpret_cast<unsigned int*>(ctx->d_keys + ASMC_MAX_BETAS + 4);  // zeroed by launch_max
    // a plain round narrows the bracket 16x, a prediction window that holds the root by much more (asmc_bisect.h): smooth
    // populations need 3 rounds whatever the tolerance.  Three are enqueued (fewer when plain rounds reach the tolerance
    // sooner); a search that is not done after them gets further rounds one at a time.
    int rounds = (int)ceil(log2((1.0 - beta0) / tol) / BIS_LEVELS - 1e-9);
    if (rounds < 1) rounds = 1;
    if (rounds > 3) rounds = 3;
    unsigned long long* hk = reinterpret_cast<unsigned long long*>(h + 40);
    int launched = 0;
    for (int attempt = 0; attempt < 32; attempt++) {
        for (; launched < rounds; launched++) {
            ASMC_LAUNCH(ctx, st, "k_bis_sums", k_bis_sums, dim3(grid), dim3(BIS_THREADS), 0, st, n, ll, lp, lq, d_st, ctx->d_partials,
                        d_ticket, launched, init, (const unsigned long long*)ctx->d_keys, (double*)nullptr);
            ASMC_LAUNCH_CHECK();
        }
        ASMC_HIP(hipMemcpyAsync(h, d_st, sizeof(double) * 40, hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipMemcpyAsync(hk, ctx->d_keys + ASMC_MAX_BETAS, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipStreamSynchronize(st));
        if (h[2] != 0.0) break;  // converged
        rounds++;
    }
    out_host[0] = h[0];               // beta_min = beta*
    out_host[1] = h[1];               // beta_max
    out_host[2] = h[2];               // converged flag
    out_host[3] = h[6];               // device rounds
    out_host[4] = h[9];               // ESS(1.0)/N
    out_host[5] = (double)hk[0];      // NaN log-weights
    out_host[6] = h[11];              // (m, S1, S2) of the log-sum-exp at beta*, valid when out[9] != 0
    out_host[7] = h[12];
    out_host[8] = h[13];
    out_host[9] = h[14];
    out_host[10] = h[10];             // (m, S1, S2) at beta = 1
    out_host[11] = h[32];
    out_host[12] = h[33];
    return ASMC_OK;
}

}  // extern "C"

int asmc_is_weights_launch(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                           double target_eff, double tol, double* w, double* tiles, double* rec, hipStream_t st) {
    if (ctx->isw_disabled) {
        asmc_set_error("asmc_importance_step: disabled on this context (an earlier launch was not fully resident)");
        return ASMC_ERR_UNSUPPORTED;
    }
    const BisInit init = {beta0, target_eff, tol, log((double)n), (double)n, bis_plain_mode()};
    int64_t chunk = ISW_CHUNK;
    int64_t grid = (n + chunk - 1) / chunk;
    ASMC_REQUIRE(ctx->num_cu <= ISW_THREADS, "more compute units than the first round's reduction has threads");
    const bool reg = grid <= ctx->num_cu;
    if (!reg) {
        chunk = (n + ctx->num_cu - 1) / ctx->num_cu;
        chunk = (chunk + ISW_CHUNK - 1) / ISW_CHUNK * ISW_CHUNK;
        grid = (n + chunk - 1) / chunk;
    }
    double* d_st = ctx->d_small + 2560;
    IswBases bases;
    bases.top = ctx->bar_base[0];
    if (getenv("ASMC_ISW_TEST_TIMEOUT")) bases.top += 1000000u;  // test hook: the barriers of this launch can never complete
    ctx->bar_base[0] += (unsigned int)ISW_BARRIERS * (unsigned int)(grid < ISW_GROUPS ? grid : ISW_GROUPS);
    for (int g = 0; g < ISW_GROUPS; g++) {
        bases.group[g] = ctx->bar_base[1 + g];
        if (g < grid) ctx->bar_base[1 + g] += (unsigned int)ISW_BARRIERS * (unsigned int)((grid - g + ISW_GROUPS - 1) / ISW_GROUPS);
    }
    if (reg) {
        static bool attr_set_dev[ASMC_MAX_DEVICES] = {false}; bool& attr_set = attr_set_dev[asmc_dev_slot(ctx)];
        if (!attr_set) {
            ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_is_weights<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ISW_CHUNK * (int)sizeof(double)));
            attr_set = true;
        }
        ASMC_LAUNCH(ctx, st, "k_is_weights", k_is_weights<true>, dim3((unsigned)grid), dim3(ISW_THREADS),
                    2 * ISW_CHUNK * sizeof(double), st, n, ll, lp, lq, d_st,
                    ctx->d_partials, ctx->d_bar, bases, init, chunk, w, tiles, rec);
    } else
        ASMC_LAUNCH(ctx, st, "k_is_weights", k_is_weights<false>, dim3((unsigned)grid), dim3(ISW_THREADS), 0, st, n, ll, lp, lq, d_st,
                    ctx->d_partials, ctx->d_bar, bases, init, chunk, w, tiles, rec);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

extern "C" {

// The read-back of asmc_importance_result put on the stream NOW, with an event behind it: a caller that enqueues more work
// behind the step (the gather's consumers: asmc_mean_gram_enqueue) can then collect the step's scalars as soon as the step
// itself is done, while that work is still running.
int asmc_importance_result_enqueue(asmc_ctx* ctx, asmc_stream stream) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    hipStream_t st = as_stream(stream);
    double* h = ctx->h_pinned + 4096 + 512;
    unsigned int* hp = reinterpret_cast<unsigned int*>(h + 64);
    ASMC_HIP(hipMemcpyAsync(h, ctx->d_small + 2560, sizeof(double) * 64, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(hp, ctx->d_bar + ISW_POISON_CELL, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    if (!ctx->ev_is) ASMC_HIP(hipEventCreateWithFlags(&ctx->ev_is, hipEventDisableTiming));
    ASMC_HIP(hipEventRecord(ctx->ev_is, st));
    ctx->is_result_pending = 1;
    return ASMC_OK;
}

int asmc_importance_result(asmc_ctx* ctx, double* out_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && out_host, "null pointer");
    hipStream_t st = as_stream(stream);
    double* h = ctx->h_pinned + 4096 + 512;
    unsigned int* hp = reinterpret_cast<unsigned int*>(h + 64);
    if (ctx->is_result_pending) {  // asmc_importance_result_enqueue has put the copies on the stream already
        ctx->is_result_pending = 0;
        ASMC_HIP(hipEventSynchronize(ctx->ev_is));
    } else {
        ASMC_HIP(hipMemcpyAsync(h, ctx->d_small + 2560, sizeof(double) * 64, hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipMemcpyAsync(hp, ctx->d_bar + ISW_POISON_CELL, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipStreamSynchronize(st));
    }
    out_host[0] = h[0], out_host[1] = h[1], out_host[2] = h[2], out_host[3] = h[6], out_host[4] = h[9];
    out_host[5] = h[15];
    out_host[6] = h[11], out_host[7] = h[12], out_host[8] = h[13], out_host[9] = h[14];
    out_host[10] = h[10], out_host[11] = h[32], out_host[12] = h[33];
    out_host[13] = h[40], out_host[14] = h[41], out_host[15] = h[45];
    if (hp[0] != 0) {
        // a grid barrier timed out (isw_barrier): nothing of this step is valid; reset the counters and leave the
        // persistent kernel alone from now on - the caller redoes the step through the step-by-step entry points
        ASMC_HIP(hipMemsetAsync(ctx->d_bar, 0, sizeof(unsigned int) * ISW_BAR_STRIDE * (ISW_GROUPS + 3), st));
        memset(ctx->bar_base, 0, sizeof(ctx->bar_base));
        ctx->isw_disabled = 1;
        out_host[2] = 0.0, out_host[9] = 0.0, out_host[15] = 0.0;
    }
#ifdef ISW_STAMP
    fprintf(stderr, "k_is_weights stamps (us, block 0):");
    for (int k = 48; k < 64; k++) fprintf(stderr, " %.2f", h[k]);
    fprintf(stderr, "\n");
#endif
    return ASMC_OK;
}

int asmc_importance_available(asmc_ctx* ctx) { return ctx && !ctx->isw_disabled ? 1 : 0; }

static inline void bis_state(asmc_ctx* ctx, double** d_st, unsigned int** d_ticket) {
    *d_st = ctx->d_small + 2560;
    *d_ticket = reinterpret_cast<unsigned int*>(ctx->d_keys + ASMC_MAX_BETAS + 4);  // zeroed by launch_max
}

int asmc_find_beta_shard_reduce(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                                double beta0, int round, double* rec_dev, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(rec_dev != nullptr && round >= 0, "null record / negative round");
    ASMC_REQUIRE(beta0 >= 0.0 && beta0 < 1.0, "bad beta0");
    hipStream_t st = as_stream(stream);
    double* d_st;
    unsigned int* d_ticket;
    bis_state(ctx, &d_st, &d_ticket);
    if (round == 0) {
        const double one = 1.0;
        rc = launch_max(ctx, n, ll, lp, lq, beta0, &one, 1, st);  // local m(1), local NaN census, ticket := 0
        if (rc) return rc;
    }
    const BisInit init = {beta0, 0.0, 0.0, 0.0, 0.0, 0.0};  // the reduction half only needs beta0
    const int grid = grid_for(n, BIS_THREADS, ctx->num_cu);
    ASMC_LAUNCH(ctx, st, "k_bis_sums", k_bis_sums, dim3(grid), dim3(BIS_THREADS), 0, st, n, ll, lp, lq, d_st, ctx->d_partials,
                d_ticket, round, init, (const unsigned long long*)ctx->d_keys, rec_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_find_beta_shard_decide(asmc_ctx* ctx, const double* recs_dev, int world, int64_t n_global, double beta0,
                                double target_eff, double tol, int round, asmc_stream stream) {
    ASMC_REQUIRE(ctx && recs_dev, "null pointer");
    ASMC_REQUIRE(world >= 1 && n_global > 0 && round >= 0, "bad world / n_global / round");
    ASMC_REQUIRE(tol > 0.0 && beta0 >= 0.0 && beta0 < 1.0, "bad beta0 / tolerance");
    hipStream_t st = as_stream(stream);
    double* d_st;
    unsigned int* d_ticket;
    bis_state(ctx, &d_st, &d_ticket);
    const BisInit init = {beta0, target_eff, tol, log((double)n_global), (double)n_global, bis_plain_mode()};
    ASMC_LAUNCH(ctx, st, "k_bis_decide", k_bis_decide, dim3(1), dim3(BIS_THREADS), 0, st, recs_dev, world, d_st, round, init);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// Rounds first .. last-1 of the sharded search as ONE call: reduce -> ncclAllGather of the ranks' records on the library's
// own communicator (asmc_set_rccl + asmc_set_rccl_allgather; same stream, no stream hop, no host callback) -> decide.
int asmc_find_beta_shard_rounds(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                                double target_eff, double tol, int world, int64_t n_global, double* rec_dev, double* recs_dev,
                                int first, int last, asmc_stream stream) {
    ASMC_REQUIRE(ctx && rec_dev && recs_dev, "null pointer");
    ASMC_REQUIRE(ctx->rccl_allgather && ctx->rccl_comm, "asmc_set_rccl / asmc_set_rccl_allgather first");
    ASMC_REQUIRE(first >= 0 && last >= first, "bad round range");
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
    const allgather_fn allgather = reinterpret_cast<allgather_fn>(ctx->rccl_allgather);
    const int nccl_f64 = 8;  // rccl.h: ncclFloat64
    for (int r = first; r < last; r++) {
        int rc = asmc_find_beta_shard_reduce(ctx, n, ll, lp, lq, beta0, r, rec_dev, stream);
        if (rc) return rc;
        if (allgather(rec_dev, recs_dev, (size_t)ASMC_BIS_REC, nccl_f64, ctx->rccl_comm, as_stream(stream)) != 0) {
            asmc_set_error("asmc_find_beta_shard_rounds: ncclAllGather failed");
            return ASMC_ERR_ARG;
        }
        rc = asmc_find_beta_shard_decide(ctx, recs_dev, world, n_global, beta0, target_eff, tol, r, stream);
        if (rc) return rc;
    }
    return ASMC_OK;
}

int asmc_find_beta_shard_result(asmc_ctx* ctx, double* out_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && out_host, "null pointer");
    hipStream_t st = as_stream(stream);
    double* h = ctx->h_pinned + 4096 + 512;
    ASMC_HIP(hipMemcpyAsync(h, ctx->d_small + 2560, sizeof(double) * 40, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    out_host[0] = h[0], out_host[1] = h[1], out_host[2] = h[2], out_host[3] = h[6], out_host[4] = h[9];
    out_host[5] = h[15];
    out_host[6] = h[11], out_host[7] = h[12], out_host[8] = h[13], out_host[9] = h[14];
    out_host[10] = h[10], out_host[11] = h[32], out_host[12] = h[33];
    return ASMC_OK;
}

int asmc_weights_m2(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                    double beta0, double beta, double m, double mean_u, double* m2_host,
                    asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(m2_host != nullptr, "null host pointer");
    hipStream_t st = as_stream(stream);
    const int grid = reduce_grid(ctx, n, 1);
    ASMC_LAUNCH(ctx, st, "k_weights_m2", k_weights_m2, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, beta0 - beta,
                       beta - beta0, m, mean_u, ctx->d_partials);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_finalize_columns", k_finalize_columns, dim3(1), dim3(64), 0, st, grid, 1, ctx->d_partials,
                       ctx->d_small, 1, 0, (const unsigned long long*)nullptr,
                       (const unsigned long long*)nullptr);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(double), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    *m2_host = ctx->h_pinned[0];
    return ASMC_OK;
}

int asmc_weights_m2_lse(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                        double beta0, double beta, double m, double mean_u, double shift, double mp,
                        double* out_host, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(out_host != nullptr, "null host pointer");
    hipStream_t st = as_stream(stream);
    const int grid = reduce_grid(ctx, n, 1);
    ASMC_LAUNCH(ctx, st, "k_weights_m2_lse", k_weights_m2_lse, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, beta0 - beta,
                beta - beta0, m, mean_u, shift, mp, ctx->d_partials);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_finalize_columns", k_finalize_columns, dim3(2), dim3(64), 0, st, grid, 2, ctx->d_partials,
                ctx->d_small, 1, 0, (const unsigned long long*)nullptr, (const unsigned long long*)nullptr);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    out_host[0] = ctx->h_pinned[0];
    out_host[1] = ctx->h_pinned[1];
    return ASMC_OK;
}

int asmc_weights_m2_lse_dev(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                            double beta0, double beta, double m, double mean_u, double shift, double mp,
                            double* out_dev, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(out_dev != nullptr, "null device pointer");
    hipStream_t st = as_stream(stream);
    const int grid = reduce_grid(ctx, n, 1);
    ASMC_LAUNCH(ctx, st, "k_weights_m2_lse", k_weights_m2_lse, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, beta0 - beta,
                beta - beta0, m, mean_u, shift, mp, ctx->d_partials);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_finalize_columns", k_finalize_columns, dim3(2), dim3(64), 0, st, grid, 2, ctx->d_partials,
                out_dev, 1, 0, (const unsigned long long*)nullptr, (const unsigned long long*)nullptr);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// Sharded importance step with device-resident scalars (include/asmc.h): the search state of asmc_find_beta_shard_decide
// stays where it is, these passes read it there.
// state record S_r (after r closed rounds) of the folded sharded search: two buffers, S_r in buffer r & 1 - the kernel that
// closes round r - 1 reads S_{r-1} while its block 0 writes S_r
static inline double* shard_state_buf(asmc_ctx* ctx, int r) { return ctx->d_small + 2560 + ((r & 1) ? 64 : 0); }

int asmc_find_beta_shard_round(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                               double target_eff, double tol, int world, int64_t n_global, int round, const double* recs_prev_dev,
                               double* rec_dev, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(rec_dev != nullptr && round >= 0 && (round == 0 || recs_prev_dev), "null record / negative round");
    ASMC_REQUIRE(world >= 1 && n_global > 0 && tol > 0.0 && beta0 >= 0.0 && beta0 < 1.0, "bad world / n_global / beta0 / tolerance");
    if (round == 0) return asmc_find_beta_shard_reduce(ctx, n, ll, lp, lq, beta0, 0, rec_dev, stream);
    hipStream_t st = as_stream(stream);
    double* d_st;
    unsigned int* d_ticket;
    bis_state(ctx, &d_st, &d_ticket);
    const BisInit init = {beta0, target_eff, tol, log((double)n_global), (double)n_global, bis_plain_mode()};
    const int grid = grid_for(n, BIS_THREADS, ctx->num_cu);
    ASMC_LAUNCH(ctx, st, "k_bis_sums", k_bis_sums, dim3(grid), dim3(BIS_THREADS), 0, st, n, ll, lp, lq, shard_state_buf(ctx, round),
                ctx->d_partials, d_ticket, round, init, (const unsigned long long*)ctx->d_keys, rec_dev, recs_prev_dev, world,
                (const double*)shard_state_buf(ctx, round - 1));
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// Sharded importance step with device-resident scalars (include/asmc.h): the search state stays on the device, these passes
// read it there.  recs_last_dev given: the search ran as asmc_find_beta_shard_round x n_rounds and its last round is closed
// inside this pass; NULL: the state asmc_find_beta_shard_decide left.
int asmc_weights_m2_lse_shard(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq, double* out_dev,
                              const double* recs_last_dev, int world, int64_t n_global, double beta0, double target_eff,
                              double tol, int n_rounds, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(out_dev != nullptr, "null device pointer");
    ASMC_REQUIRE(!recs_last_dev || (world >= 1 && n_global > 0 && n_rounds >= 1 && tol > 0.0 && beta0 >= 0.0 && beta0 < 1.0),
                 "bad world / n_global / rounds / beta0 / tolerance");
    hipStream_t st = as_stream(stream);
    double* d_st;
    unsigned int* d_ticket;
    bis_state(ctx, &d_st, &d_ticket);
    const int grid = reduce_grid(ctx, n, 1);
    BisInit init = {beta0, target_eff, tol, 0.0, 0.0, bis_plain_mode()};
    const double* st_in = d_st;
    double* st_out = nullptr;
    if (recs_last_dev) {
        init.logN = log((double)n_global), init.N = (double)n_global;
        st_in = shard_state_buf(ctx, n_rounds - 1);
        st_out = shard_state_buf(ctx, n_rounds);
    }
    ctx->shard_st = recs_last_dev ? st_out : d_st;
    ASMC_LAUNCH(ctx, st, "k_weights_m2_lse_shard", k_weights_m2_lse_shard, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, st_in,
                ctx->d_partials, ctx->d_bar + SHARD_TICKET_CELL, out_dev, recs_last_dev, world, n_rounds - 1, init, st_out);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_normalized_weights_shard(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                                  const double* parts_dev, int world, int rank, double carry_uniform, double* w_out,
                                  double* carry_out_dev, double* tile_sums_dev, double* state_copy_dev, int pack_records,
                                  asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(parts_dev && w_out && carry_out_dev && tile_sums_dev, "null pointer");
    ASMC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    ASMC_REQUIRE(carry_uniform >= 0.0 && carry_uniform < 1.0, "bad uniform carry");
    ASMC_REQUIRE(ctx->shard_st != nullptr, "asmc_weights_m2_lse_shard first");
    hipStream_t st = as_stream(stream);
    const int64_t n_tiles = (n + ASMC_SCAN_TILE - 1) / ASMC_SCAN_TILE;
    ASMC_LAUNCH(ctx, st, "k_weights_map_shard", k_weights_map_shard, dim3((unsigned)n_tiles), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq,
                (const double*)ctx->shard_st, parts_dev, world, rank, carry_uniform, w_out, carry_out_dev, tile_sums_dev, state_copy_dev,
                pack_records ? ctx->d_rec : (double*)nullptr);
    ASMC_LAUNCH_CHECK();
    if (pack_records) {  // d_rec rewritten: a new generation, held for asmc_rec_claim
        ctx->rec_token++;
        ctx->rec_hold_src[0] = ll, ctx->rec_hold_src[1] = lp, ctx->rec_hold_src[2] = lq;
        ctx->rec_hold_n = n, ctx->rec_hold_token = ctx->rec_token;
    }
    return ASMC_OK;
}

// The one synchronisation of the sharded importance step: res_dev = {search state [40] (asmc_normalized_weights_shard's
// state_copy_dev), the ranks' (m2, S1') pairs [2 world], their (kept, fail) int64 pairs [2 world]}, one contiguous buffer ->
// out_host[13 + 4 world] (asmc_find_beta_shard_result's 13 values, the pairs, the int64 pairs as doubles).
int asmc_shard_step_result(asmc_ctx* ctx, const double* res_dev, int world, double* out_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && res_dev && out_host, "null pointer");
    ASMC_REQUIRE(world >= 1 && world <= 64, "world out of range");
    hipStream_t st = as_stream(stream);
    double* h = ctx->h_pinned + 4096 + 512;
    ASMC_HIP(hipMemcpyAsync(h, res_dev, sizeof(double) * (40 + 4 * world), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    out_host[0] = h[0], out_host[1] = h[1], out_host[2] = h[2], out_host[3] = h[6], out_host[4] = h[9];
    out_host[5] = h[15];
    out_host[6] = h[11], out_host[7] = h[12], out_host[8] = h[13], out_host[9] = h[14];
    out_host[10] = h[10], out_host[11] = h[32], out_host[12] = h[33];
    for (int i = 0; i < 2 * world; i++) out_host[13 + i] = h[40 + i];
    const int64_t* hi = reinterpret_cast<const int64_t*>(h + 40 + 2 * world);
    for (int i = 0; i < 2 * world; i++) out_host[13 + 2 * world + i] = (double)hi[i];
    return ASMC_OK;
}

int asmc_log_weights(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp, const double* lq,
                     double beta0, double beta, double shift, double* lw_out, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(lw_out != nullptr, "null output");
    const int grid = grid_for(n, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, as_stream(stream), "k_weights_map<0>", k_weights_map<0>, dim3(grid), dim3(ASMC_BLOCK), 0, as_stream(stream), n, ll, lp, lq,
                       beta0 - beta, beta - beta0, shift, 0.0, lw_out);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_normalized_weights(asmc_ctx* ctx, int64_t n, const double* ll, const double* lp,
                            const double* lq, double beta0, double beta, double shift, double lse,
                            double* w_out, asmc_stream stream) {
    int rc = check_common(ctx, n, ll, lp, lq);
    if (rc) return rc;
    ASMC_REQUIRE(w_out != nullptr, "null output");
    const int grid = grid_for(n, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, as_stream(stream), "k_weights_map<1>", k_weights_map<1>, dim3(grid), dim3(ASMC_BLOCK), 0, as_stream(stream), n, ll, lp, lq,
                       beta0 - beta, beta - beta0, shift, lse, w_out);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

}  // extern "C"

// NaN / inf counts of v[0..n), enqueued only (the mutation calls read them back with their own results); the counts land in
// asmc_count_slot(ctx) - the slot of the LAST enqueue
static inline unsigned long long* count_slot(asmc_ctx* ctx, unsigned gen) {
    return reinterpret_cast<unsigned long long*>(ctx->d_bar + COUNT_CELLS) + 2 * (gen & 1u);
}
unsigned long long* asmc_count_slot(asmc_ctx* ctx) { return count_slot(ctx, ctx->count_gen); }
int asmc_count_nonfinite_enqueue(asmc_ctx* ctx, int64_t n, const double* v, hipStream_t st) {
    ctx->count_gen++;
    const int grid = grid_for(n, ASMC_BLOCK * 8, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, st, "k_count_nonfinite", k_count_nonfinite, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, v,
                count_slot(ctx, ctx->count_gen), count_slot(ctx, ctx->count_gen + 1));
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

extern "C" {
int asmc_count_nonfinite(asmc_ctx* ctx, int64_t n, const double* v, int64_t* n_nan_host,
                         int64_t* n_inf_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && v, "null pointer");
    ASMC_REQUIRE(n > 0, "n must be positive");
    hipStream_t st = as_stream(stream);
    const int rc = asmc_count_nonfinite_enqueue(ctx, n, v, st);
    if (rc) return rc;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(ctx->h_pinned);
    ASMC_HIP(hipMemcpyAsync(h, asmc_count_slot(ctx), sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    if (n_nan_host) *n_nan_host = (int64_t)h[0];
    if (n_inf_host) *n_inf_host = (int64_t)h[1];
    return ASMC_OK;
}

}  // extern "C"
