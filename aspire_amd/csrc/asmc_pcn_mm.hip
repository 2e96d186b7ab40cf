// asmc_pcn_mm.hip — pCN mutation for d = 64 / 128 on the fp64 matrix cores (BASELINE config 5: d = 128 mixture target).
//
// Same specification as asmc_pcn.hip (reference seam src/aspire/samplers/smc/minipcn.py:69-135,
// tempered target src/aspire/samplers/smc/base.py:507-519), whitened-state stepping:
//   MM_WHITEN    y = Linv (x - mu)                 once per mutation
//   MM_STEP      y' = a y + rho xi;  x' = mu + L y';  densities at x';  accept y <- y'
//   MM_UNWHITEN  x = mu + L y;  ll / lp / lq re-evaluated at the stored x
// At d = 128 a triangular mat-vec is 8256 FMAs per particle with 66 KB of coefficients: far too many for the
// scalar-operand scheme of the d <= 32 kernels, and exactly GEMM shaped:  X'^T [d x M] = L [d x d] Y'^T [d x M].
// It runs on v_mfma_f64_16x16x4_f64 with the coefficient matrix as the A operand (one f64 per lane per
// instruction, read from an LDS image packed in (row block, k-step, lane) order, zero blocks above the diagonal
// skipped) and the particles' coordinates as the B operand.  Lane l = (p = l & 15, h = l >> 4) supplies
// B[k = h][j = p], so it keeps the coordinates 4 s + h (s = 0..d/4-1) of particle p; the result block has column
// p on the lane and rows h + 4 r — the SAME coordinates — so noise, mu, the densities' quadratic forms and the
// write-back are all lane-local, and only the per-particle scalars are summed over the four lanes p + 16 h.
// A wave steps 16 particles at a time; a block keeps the 72 KB (d = 128) operand image in LDS and loops over
// particle groups, two blocks per CU.
#include <stdlib.h>

#include "asmc_common.h"
#include "asmc_pcn_dev.h"

typedef double doublex4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int mm_ksum(int nb) { return 2 * nb * (nb + 1); }  // sum over row blocks of (4 ib + 4)

// pack a row-major [d x d] lower-triangular matrix into MFMA A-operand order
__global__ __launch_bounds__(256) void k_mm_pack(int d, const double* __restrict__ M, double* __restrict__ pack) {
    const int nb = d / 16;
    const int total = mm_ksum(nb) * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, ks = e >> 6;
        int ib = 0;
        while (mm_ksum(ib + 1) <= ks) ib++;
        const int s = ks - mm_ksum(ib);
        const int i = 16 * ib + (lane & 15), k = 4 * s + (lane >> 4);
        pack[e] = (k <= i) ? M[(size_t)i * d + k] : 0.0;
    }
}

// out = M v for this lane's coordinates (out[4 ib + r] = coordinate 16 ib + 4 r + h).  Row blocks are taken in pairs
// (ib, NB-1-ib): equal work per pair, and consecutive MFMAs alternate between two accumulators.
template <int D>
__device__ __forceinline__ void mm_trimatvec(const double* __restrict__ sA, const double (&v)[D / 4], double (&out)[D / 4],
                                             int lane) {
    constexpr int NB = D / 16;
#pragma unroll
    for (int pr = 0; pr < NB / 2; pr++) {
        const int ia = pr, ib = NB - 1 - pr;
        doublex4 acc_a = {0.0, 0.0, 0.0, 0.0}, acc_b = {0.0, 0.0, 0.0, 0.0};
        const double* Aa = sA + (size_t)mm_ksum(ia) * 64 + lane;
        const double* Ab = sA + (size_t)mm_ksum(ib) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 4 * ib + 4; s++) {
            if (s < 4 * ia + 4) acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(Aa[(size_t)s * 64], v[s], acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[(size_t)s * 64], v[s], acc_b, 0, 0, 0);
            // keep the scheduler from hoisting all 144 operand reads to the top (it spills 270 VGPRs otherwise)
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            out[4 * ia + r] = acc_a[r];
            out[4 * ib + r] = acc_b[r];
        }
    }
}

__device__ __forceinline__ double quad_sum(double q) {  // over the four lanes p, p+16, p+32, p+48 of a particle
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    return q;
}

// diagonal-mixture log-density of the particle whose coordinates 4 s + h this lane holds (same formula as
// mixture_eval in asmc_pcn.hip: log-sum-exp over the components' logw - q / 2)
template <int D>
__device__ __forceinline__ double mm_mixture(const MixDev& m, const double (&xv)[D / 4], int h) {
    double terms[ASMC_MAX_COMPONENTS];
    double best = -INFINITY;
    for (int c = 0; c < m.C; c++) {
        const double* mu = m.mu + (size_t)c * D + h;
        const double* pr = m.prec + (size_t)c * D + h;
        double q = 0.0;
#pragma unroll
        for (int s = 0; s < D / 4; s++) {
            const double t = xv[s] - mu[4 * s];
            q = fma(t * t, pr[4 * s], q);
        }
        q = quad_sum(q);
        terms[c] = m.logw[c] - 0.5 * q;
        best = fmax(best, terms[c]);
    }
    if (m.C == 1) return terms[0];
    if (!(best > -INFINITY)) return best;
    double sum = 0.0;
    for (int c = 0; c < m.C; c++) sum += exp(terms[c] - best);
    return best + log(sum);
}

template <typename T, int D, int NOISE, int MODE>
__global__ __launch_bounds__(256, 2) void k_pcn_mm(int64_t n, T* __restrict__ x, double* __restrict__ ll,
                                                  double* __restrict__ lp, double* __restrict__ lq,
                                                  const double* __restrict__ pack, PcnDev p,
                                                  const double* __restrict__ rho_ptr, uint32_t step,
                                                  long long* __restrict__ block_counts) {
    extern __shared__ __align__(16) double sA[];
    constexpr int KS = D / 4;
    constexpr int TOTAL = mm_ksum(D / 16) * 64;
    for (int e = threadIdx.x * 2; e < TOTAL; e += 256 * 2)
        *reinterpret_cast<double2*>(sA + e) = *reinterpret_cast<const double2*>(pack + e);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pp = lane & 15, h = lane >> 4;
    const double rho = (MODE == MM_STEP) ? *rho_ptr : 0.0;
    const double a = sqrt(1.0 - rho * rho);
    const int64_t n_groups = (n + 15) / 16;
    long long n_acc = 0;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < n_groups; g += (int64_t)gridDim.x * 4) {
        const int64_t row = g * 16 + pp;
        const bool valid = row < n;
        T* xr = x + row * D + h;
        double v[KS], o[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) v[s] = valid ? (double)xr[4 * s] : 0.0;
        if (MODE == MM_WHITEN) {
#pragma unroll
            for (int s = 0; s < KS; s++) v[s] -= p.mu[4 * s + h];
            mm_trimatvec<D>(sA, v, o, lane);
            if (valid) {
#pragma unroll
                for (int s = 0; s < KS; s++) xr[4 * s] = (T)o[s];
            }
        } else if (MODE == MM_UNWHITEN) {
            mm_trimatvec<D>(sA, v, o, lane);
#pragma unroll
            for (int s = 0; s < KS; s++) o[s] = (double)(T)(p.mu[4 * s + h] + o[s]);
            const double nll = mm_mixture<D>(p.ll, o, h), nlp = mm_mixture<D>(p.lp, o, h), nlq = mm_mixture<D>(p.lq, o, h);
            if (valid) {
#pragma unroll
                for (int s = 0; s < KS; s++) xr[4 * s] = (T)o[s];
                if (h == 0) ll[row] = nll, lp[row] = nlp, lq[row] = nlq;
            }
        } else {
            const unsigned long long gid = p.gid0 + (unsigned long long)row;
            double q0 = 0.0, q1 = 0.0;
#pragma unroll
            for (int s = 0; s < KS; s++) q0 = fma(v[s], v[s], q0);
            q0 = quad_sum(q0);
#pragma unroll
            for (int s = 0; s < KS; s++) {
                double z;
                if (NOISE == ASMC_NOISE_F32) {  // coordinates 4 s .. 4 s + 3 come from Philox block s
                    double z0, z1, z2, z3;
                    normal_quad_f32(p.seed, gid, step, (uint32_t)s, z0, z1, z2, z3);
                    z = h == 0 ? z0 : (h == 1 ? z1 : (h == 2 ? z2 : z3));
                } else {  // coordinates 2 pr, 2 pr + 1 from pair pr
                    double z0, z1;
                    normal_pair(p.seed, gid, step, (uint32_t)(2 * s + (h >> 1)), z0, z1);
                    z = (h & 1) ? z1 : z0;
                }
                v[s] = (double)(T)fma(rho, z, a * v[s]);
                q1 = fma(v[s], v[s], q1);
            }
            q1 = quad_sum(q1);
            mm_trimatvec<D>(sA, v, o, lane);
#pragma unroll
            for (int s = 0; s < KS; s++) o[s] = (double)(T)(p.mu[4 * s + h] + o[s]);
            const double nll = mm_mixture<D>(p.ll, o, h), nlp = mm_mixture<D>(p.lp, o, h), nlq = mm_mixture<D>(p.lq, o, h);
            if (valid) {
                const double lpn = log_p_t(nll, nlp, nlq, p.beta);
                const double lpo = log_p_t(ll[row], lp[row], lq[row], p.beta);
                const double log_a = (lpn + 0.5 * q1) - (lpo + 0.5 * q0);
                const double u = accept_uniform(p.seed, gid, step);
                if (log(u) < log_a) {
#pragma unroll
                    for (int s = 0; s < KS; s++) xr[4 * s] = (T)v[s];
                    if (h == 0) {
                        ll[row] = nll, lp[row] = nlp, lq[row] = nlq;
                        n_acc++;
                    }
                }
            }
        }
    }
    if (MODE == MM_STEP) {
        __shared__ long long s_cnt[4];
        n_acc = wave_sum_ll(n_acc);
        if (lane == 0) s_cnt[wave] = n_acc;
        __syncthreads();
        if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int D, int NOISE, int MODE>
static int launch_mm(asmc_ctx* ctx, int64_t n, T* x, double* ll, double* lp, double* lq, const double* pack, const PcnDev& pd,
                     const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out, hipStream_t st) {
    const size_t lds = (size_t)mm_ksum(D / 16) * 64 * sizeof(double);
    auto kern = k_pcn_mm<T, D, NOISE, MODE>;
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int per_cu = lds > 80 * 1024 ? 1 : (lds > 40 * 1024 ? 2 : 4);
    const int64_t n_groups = (n + 15) / 16;
    const int64_t want = (n_groups + 3) / 4;
    int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    if (grid > ASMC_MAX_BLOCKS) grid = ASMC_MAX_BLOCKS;
    *grid_out = grid;
    ASMC_LAUNCH(ctx, st, MODE == MM_STEP ? "k_pcn_mm_step" : MODE == MM_WHITEN ? "k_pcn_mm_whiten" : "k_pcn_mm_unwhiten", kern,
                dim3(grid), dim3(256), lds, st, n, x, ll, lp, lq, pack, pd, rho_ptr, step, block_counts);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

bool asmc_pcn_mm_supported(int d, const void* x) { return (d == 64 || d == 128) && ((uintptr_t)x % 16 == 0); }

// packs L and Linv into ctx->d_mmtab (two images) and enqueues one launch of the requested mode
int asmc_pcn_mm_pack(asmc_ctx* ctx, const PcnDev& pd, hipStream_t st) {
    ASMC_REQUIRE(ctx->d_mmtab != nullptr, "ctx was created with d_max < 64");
    const int total = mm_ksum(pd.d / 16) * 64;
    const int grid = (total + 255) / 256;
    ASMC_LAUNCH(ctx, st, "k_mm_pack", k_mm_pack, dim3(grid), dim3(256), 0, st, pd.d, pd.L, ctx->d_mmtab);
    ASMC_LAUNCH(ctx, st, "k_mm_pack", k_mm_pack, dim3(grid), dim3(256), 0, st, pd.d, pd.Linv, ctx->d_mmtab + total);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_pcn_mm_launch(asmc_ctx* ctx, int64_t n, int x_dtype, void* x, double* ll, double* lp, double* lq, const PcnDev& pd,
                       int mode, const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                       hipStream_t st) {
    const int total = mm_ksum(pd.d / 16) * 64;
    const double* pack = (mode == MM_WHITEN) ? ctx->d_mmtab + total : ctx->d_mmtab;
#define MM_CASE(TT, DD, NZ, MD) \
    return launch_mm<TT, DD, NZ, MD>(ctx, n, (TT*)x, ll, lp, lq, pack, pd, rho_ptr, step, block_counts, grid_out, st);
#define MM_MODES(TT, DD)                                                         \
    if (mode == MM_WHITEN) { MM_CASE(TT, DD, ASMC_NOISE_F64, MM_WHITEN) }        \
    if (mode == MM_UNWHITEN) { MM_CASE(TT, DD, ASMC_NOISE_F64, MM_UNWHITEN) }    \
    if (pd.noise == ASMC_NOISE_F32) { MM_CASE(TT, DD, ASMC_NOISE_F32, MM_STEP) } \
    MM_CASE(TT, DD, ASMC_NOISE_F64, MM_STEP)
    if (x_dtype == ASMC_F64) {
        if (pd.d == 128) { MM_MODES(double, 128) }
        MM_MODES(double, 64)
    }
    if (pd.d == 128) { MM_MODES(float, 128) }
    MM_MODES(float, 64)
#undef MM_MODES
#undef MM_CASE
}
