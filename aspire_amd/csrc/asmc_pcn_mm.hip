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
// B[k = h][j = p] and receives result rows h + 4 r of column p.  The contraction order inside a k-step and the row
// order inside a block are free (they are folded into the operand image), so both are chosen such that lane (p, h)
// owns the SAME coordinates of particle p on the input and on the output side, in adjacent pairs:
//     coordinate(s, h) = 8 (s / 2) + 2 h + s % 2,   s = 0 .. d/4 - 1
// Noise (one Philox pair = one owned pair of coordinates, no redundant draws), mu, the densities' quadratic forms
// and the write-back are then lane-local with 16-byte accesses, and only per-particle scalars are summed over the
// four lanes p + 16 h.  A wave steps 16 particles at a time; a block of 8 waves keeps the 72 KB (d = 128) operand
// image and the densities' tables in LDS and loops over particle groups, one block per CU.
#include <stdlib.h>

#include "asmc_common.h"
#include "asmc_pcn_dev.h"

typedef double doublex4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int mm_ksum(int nb) { return 2 * nb * (nb + 1); }  // sum over row blocks of (4 ib + 4)

__host__ __device__ constexpr int mm_coord(int s, int h) { return 8 * (s / 2) + 2 * h + (s % 2); }

// pack a row-major [d x d] lower-triangular matrix into MFMA A-operand order
__global__ __launch_bounds__(256) void k_mm_pack(int d, const double* __restrict__ M, double* __restrict__ pack) {
    const int nb = d / 16;
    const int total = mm_ksum(nb) * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, ks = e >> 6;
        int ib = 0;
        while (mm_ksum(ib + 1) <= ks) ib++;
        const int s = ks - mm_ksum(ib);
        // A-operand lane: in-block result row rho = lane & 15 (result lane h = rho % 4, register r = rho / 4), k = lane >> 4
        const int rho = lane & 15, kk = lane >> 4;
        const int i = mm_coord(4 * ib + rho / 4, rho % 4), k = mm_coord(s, kk);
        pack[e] = (k <= i) ? M[(size_t)i * d + k] : 0.0;
    }
}

// out = M v for this lane's coordinates (out[4 ib + r] = coordinate mm_coord(4 ib + r, h)).  Row blocks are taken in pairs
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

// Density tables in LDS, lane order: tab[c][s / 2][h] = {mu(s), mu(s+1), prec(s), prec(s+1)} for even s (32 bytes),
// so a lane fetches the constants of an owned coordinate pair with two 16-byte reads.
template <int D>
__device__ __forceinline__ void mm_stage_tables(double* tab, const MixDev& m, int tid, int nthreads) {
    const int per_c = D * 2;  // doubles per component
    for (int e = tid; e < m.C * per_c; e += nthreads) {
        const int c = e / per_c, w = e - c * per_c;
        const int q = w & 3, hh = (w >> 2) & 3, sp = w >> 4;  // w = (sp * 4 + hh) * 4 + q
        const int coord = mm_coord(2 * sp + (q & 1), hh);
        tab[e] = (q < 2) ? m.mu[(size_t)c * D + coord] : m.prec[(size_t)c * D + coord];
    }
}

// diagonal-mixture log-density of the particle whose coordinates this lane holds (same formula as mixture_eval in
// asmc_pcn.hip: log-sum-exp over the components' logw - q / 2)
template <int D>
__device__ __forceinline__ double mm_mixture(const MixDev& m, const double* __restrict__ tab, const double* __restrict__ logw,
                                             const double (&xv)[D / 4], int h) {  // logw: the components' log-weights, STAGED IN LDS by the caller
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
        q = quad_sum(q);
        // (read through the mixture's device pointer this was a global load with a full `s_waitcnt vmcnt(0)` per component and
        // density, inside the loop over the groups: the wave drained its row loads and stores three times per group - round 6)
        terms[c] = logw[c] - 0.5 * q;
        best = fmax(best, terms[c]);
    }
    if (m.C == 1) return terms[0];
    if (!(best > -INFINITY)) return best;
    double sum = 0.0;
    for (int c = 0; c < m.C; c++) sum += exp(terms[c] - best);
    return best + log(sum);
}

#define MM_THREADS 512
#define MM_WAVES (MM_THREADS / 64)

template <typename T, int D, int NOISE, int MODE>
__global__ __launch_bounds__(MM_THREADS) void k_pcn_mm(int64_t n, T* __restrict__ x, double* __restrict__ ll,
                                                      double* __restrict__ lp, double* __restrict__ lq,
                                                      const double* __restrict__ pack, PcnDev p,
                                                      const double* __restrict__ rho_ptr, uint32_t step,
                                                      long long* __restrict__ block_counts) {
    extern __shared__ __align__(16) double smem[];
    constexpr bool TP = MODE == MM_STEP_T || MODE == MM_XPROPOSE_T;
    constexpr int M = MODE == MM_STEP_T ? MM_STEP : MODE == MM_XPROPOSE_T ? MM_XPROPOSE : MODE;
    // MM_XPROPOSE (the split path's proposal half, arbitrary callables as densities): both operand images are resident -
    // `pack` = L's image, followed by Linv's; ll / lp receive the two quadratic forms, lq is the x' buffer (rows of T)
    constexpr int KS = D / 4;
    constexpr int TOTAL = mm_ksum(D / 16) * 64;
    double* sA = smem;
    double* sB = sA + TOTAL;                   // MM_XPROPOSE: the image of Linv behind L's
    double* s_mu = sA + (M == MM_XPROPOSE ? 2 : 1) * TOTAL;  // [D / 8][4 h][2]  reference mean in lane order
    double* t_ll = s_mu + D;                   // density tables (MM_WHITEN does not need them)
    double* t_lp = t_ll + (size_t)p.ll.C * D * 2;
    double* t_lq = t_lp + (size_t)p.lp.C * D * 2;
    for (int e = threadIdx.x * 2; e < (M == MM_XPROPOSE ? 2 : 1) * TOTAL; e += MM_THREADS * 2)
        *reinterpret_cast<double2*>(sA + e) = *reinterpret_cast<const double2*>(pack + e);
    for (int e = threadIdx.x; e < D; e += MM_THREADS) {
        const int q = e & 1, hh = (e >> 1) & 3, sp = e >> 3;
        s_mu[e] = p.mu[mm_coord(2 * sp + q, hh)];
    }
    __shared__ double s_logw[3 * ASMC_MAX_COMPONENTS];  // log-weights of (ll, lp, lq)
    if (M != MM_WHITEN && M != MM_XPROPOSE && M != MM_UNWHITEN_X) {
        if (threadIdx.x < 3 * ASMC_MAX_COMPONENTS) {
            const int t = threadIdx.x / ASMC_MAX_COMPONENTS, c = threadIdx.x % ASMC_MAX_COMPONENTS;
            const MixDev& mt = t == 0 ? p.ll : t == 1 ? p.lp : p.lq;
            s_logw[threadIdx.x] = c < mt.C ? mt.logw[c] : 0.0;
        }
        mm_stage_tables<D>(t_ll, p.ll, threadIdx.x, MM_THREADS);
        mm_stage_tables<D>(t_lp, p.lp, threadIdx.x, MM_THREADS);
        mm_stage_tables<D>(t_lq, p.lq, threadIdx.x, MM_THREADS);
    }
    bm_d2* bmt = nullptr;  // Box-Muller tables of the default noise
    if constexpr ((M == MM_STEP || M == MM_XPROPOSE) && NOISE == ASMC_NOISE_F64) {
        bmt = bm_lds();
        bm_tab_stage<MM_THREADS>(bmt, p.bmtab);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pp = lane & 15, h = lane >> 4;
    const int dn = (p.d_noise > 0 && p.d_noise < D) ? p.d_noise : D;  // real dimension of a zero-padded problem
    const double rho = (M == MM_STEP || M == MM_XPROPOSE) ? *rho_ptr : 0.0;
    const double a = sqrt(1.0 - rho * rho);
    const int64_t n_groups = (n + 15) / 16;
    long long n_acc = 0;
    struct alignas(2 * sizeof(T)) Pair {
        T a, b;
    };
    for (int64_t g = (int64_t)blockIdx.x * MM_WAVES + wave; g < n_groups; g += (int64_t)gridDim.x * MM_WAVES) {
        const int64_t row = g * 16 + pp;
        const bool valid = row < n;
        Pair* xr = reinterpret_cast<Pair*>(x + row * D + 2 * h);  // owned pairs sit 8 elements apart
        double v[KS], o[KS];
#pragma unroll
        for (int sp = 0; sp < KS / 2; sp++) {
            Pair t = {(T)0, (T)0};
            if (valid) t = xr[sp * 4];
            v[2 * sp] = (double)t.a;
            v[2 * sp + 1] = (double)t.b;
        }
        const double* my_mu = s_mu + h * 2;  // {mu(2 sp), mu(2 sp + 1)} at my_mu[8 sp]
        auto store_row = [&](const double (&w)[KS]) {
#pragma unroll
            for (int sp = 0; sp < KS / 2; sp++) {
                Pair t;
                t.a = (T)w[2 * sp];
                t.b = (T)w[2 * sp + 1];
                xr[sp * 4] = t;
            }
        };
        if (M == MM_WHITEN) {
#pragma unroll
            for (int sp = 0; sp < KS / 2; sp++) {
                const double2 m2 = *reinterpret_cast<const double2*>(my_mu + sp * 8);
                v[2 * sp] -= m2.x;
                v[2 * sp + 1] -= m2.y;
            }
            mm_trimatvec<D>(sA, v, o, lane);
            if (valid) store_row(o);
        } else if (M == MM_UNWHITEN_X) {
            mm_trimatvec<D>(sA, v, o, lane);
#pragma unroll
            for (int sp = 0; sp < KS / 2; sp++) {
                const double2 m2 = *reinterpret_cast<const double2*>(my_mu + sp * 8);
                o[2 * sp] = (double)(T)(m2.x + o[2 * sp]);
                o[2 * sp + 1] = (double)(T)(m2.y + o[2 * sp + 1]);
            }
            if (valid) store_row(o);
        } else if (M == MM_UNWHITEN) {
            mm_trimatvec<D>(sA, v, o, lane);
#pragma unroll
            for (int sp = 0; sp < KS / 2; sp++) {
                const double2 m2 = *reinterpret_cast<const double2*>(my_mu + sp * 8);
                o[2 * sp] = (double)(T)(m2.x + o[2 * sp]);
                o[2 * sp + 1] = (double)(T)(m2.y + o[2 * sp + 1]);
            }
            const double nll = mm_mixture<D>(p.ll, t_ll, s_logw, o, h), nlp = mm_mixture<D>(p.lp, t_lp, s_logw + ASMC_MAX_COMPONENTS, o, h),
                         nlq = mm_mixture<D>(p.lq, t_lq, s_logw + 2 * ASMC_MAX_COMPONENTS, o, h);
            if (valid) {
                store_row(o);
                if (h == 0) ll[row] = nll, lp[row] = nlp, lq[row] = nlq;
            }
        } else {
            const unsigned long long gid = p.gid0 + (unsigned long long)row;
            if (M == MM_XPROPOSE) {  // x -> y = Linv (x - mu)
#pragma unroll
                for (int sp = 0; sp < KS / 2; sp++) {
                    const double2 m2 = *reinterpret_cast<const double2*>(my_mu + sp * 8);
                    v[2 * sp] -= m2.x;
                    v[2 * sp + 1] -= m2.y;
                }
                mm_trimatvec<D>(sB, v, o, lane);
#pragma unroll
                for (int s = 0; s < KS; s++) v[s] = o[s];
            }
            double q0 = 0.0, q1 = 0.0;
#pragma unroll
            for (int s = 0; s < KS; s++) q0 = fma(v[s], v[s], q0);
            q0 = quad_sum(q0);
            const double rs = tpcn_scale_ct<TP>(rho, p.nu, q0, p.gam, valid ? row : 0);  // one variate per particle
            // fast-noise mode: coordinates 4 q .. 4 q + 3 come from Philox block q = 2 sp + (h >> 1); this lane owns the first
            // two of them (h even) or the last two (h odd), its partner lane 16 away the others.  One block per lane PAIR:
            // the even lane draws the block of sp = 2 m, the odd lane the block of sp = 2 m + 1, and two v_permlane16_swap
            // (odd rows of the first operand <-> even rows of the second) hand each lane its halves of both blocks.
            float zf[NOISE == ASMC_NOISE_F32 ? KS : 1];
            if (NOISE == ASMC_NOISE_F32) {
#pragma unroll
                for (int m = 0; m < KS / 4; m++) {
                    const uint32_t slot = (uint32_t)(2 * (2 * m + (h & 1)) + (h >> 1));
                    float f0, f1, f2, f3;
                    normal_quad_f32_raw(p.seed, gid, step, slot, f0, f1, f2, f3);
                    const auto s02 = __builtin_amdgcn_permlane16_swap(__float_as_uint(f0), __float_as_uint(f2), false, false);
                    const auto s13 = __builtin_amdgcn_permlane16_swap(__float_as_uint(f1), __float_as_uint(f3), false, false);
                    zf[4 * m] = __uint_as_float(s02[0]);      // sp = 2 m:     first / second owned coordinate
                    zf[4 * m + 1] = __uint_as_float(s13[0]);
                    zf[4 * m + 2] = __uint_as_float(s02[1]);  // sp = 2 m + 1
                    zf[4 * m + 3] = __uint_as_float(s13[1]);
                }
            }
            // default noise: the same sharing - a lane's pair 4 sp + h is one HALF of Philox block 2 sp + (h >> 1), whose other half
            // belongs to the partner lane 16 away.  The even lane draws the block of sp = 2 m, the odd lane that of sp = 2 m + 1
            // (all four normals each), and four v_permlane16_swap (the two words of two doubles) trade the halves: afterwards
            // every lane holds its pair of sp = 2 m in (za0, za1) and of sp = 2 m + 1 in (zb0, zb1).  Same normals as one block
            // per lane and pair (round 2), half the Philox blocks.
            auto swap64 = [](double& x, double& y) {  // odd rows of x <-> even rows of y
                const unsigned long long xb = (unsigned long long)__double_as_longlong(x), yb = (unsigned long long)__double_as_longlong(y);
                const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
                x = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
                y = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
            };
#pragma unroll
            for (int m = 0; m < KS / 4; m++) {
                double zz[4];
                if (NOISE == ASMC_NOISE_F32) {
#pragma unroll
                    for (int e = 0; e < 4; e++) zz[e] = (double)zf[4 * m + e];
                } else {
                    const uint32_t blk = (uint32_t)(2 * (2 * m + (h & 1)) + (h >> 1));
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
            }
            if (dn < D) {  // a zero-padded problem (asmc_pcn_mutate): the padded coordinates carry no noise, y' = 0 there
                q1 = 0.0;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    v[s] = mm_coord(s, h) < dn ? v[s] : 0.0;
                    q1 = fma(v[s], v[s], q1);
                }
            }
            q1 = quad_sum(q1);
            mm_trimatvec<D>(sA, v, o, lane);
#pragma unroll
            for (int sp = 0; sp < KS / 2; sp++) {
                const double2 m2 = *reinterpret_cast<const double2*>(my_mu + sp * 8);
                o[2 * sp] = (double)(T)(m2.x + o[2 * sp]);
                o[2 * sp + 1] = (double)(T)(m2.y + o[2 * sp + 1]);
            }
            if (M == MM_XPROPOSE) {  // x' and the two forms the accept kernel adds half of (asmc_pcn_accept)
                if (valid) {
                    Pair* xo = reinterpret_cast<Pair*>(reinterpret_cast<T*>(lq) + row * D + 2 * h);
#pragma unroll
                    for (int sp = 0; sp < KS / 2; sp++) {
                        Pair t;
                        t.a = (T)o[2 * sp];
                        t.b = (T)o[2 * sp + 1];
                        xo[sp * 4] = t;
                    }
                    if (h == 0) {
                        ll[row] = 2.0 * ref_corr_ct<TP>(q0, p.nu, dn);
                        lp[row] = 2.0 * ref_corr_ct<TP>(q1, p.nu, dn);
                    }
                }
                continue;
            }
            const double nll = mm_mixture<D>(p.ll, t_ll, s_logw, o, h), nlp = mm_mixture<D>(p.lp, t_lp, s_logw + ASMC_MAX_COMPONENTS, o, h),
                         nlq = mm_mixture<D>(p.lq, t_lq, s_logw + 2 * ASMC_MAX_COMPONENTS, o, h);
            if (valid) {
                const double lpn = log_p_t(nll, nlp, nlq, p.beta);
                const double lpo = log_p_t(ll[row], lp[row], lq[row], p.beta);
                const double log_a = (lpn + ref_corr_ct<TP>(q1, p.nu, dn)) - (lpo + ref_corr_ct<TP>(q0, p.nu, dn));
                const double u = accept_uniform(p.seed, gid, step);
                if (log(u) < log_a) {
                    store_row(v);
                    if (h == 0) {
                        ll[row] = nll, lp[row] = nlp, lq[row] = nlq;
                        n_acc++;
                    }
                }
            }
        }
    }
    if (M == MM_STEP) {
        __shared__ long long s_cnt[MM_WAVES];
        n_acc = wave_sum_ll(n_acc);
        if (lane == 0) s_cnt[wave] = n_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long t = 0;
            for (int w = 0; w < MM_WAVES; w++) t += s_cnt[w];
            block_counts[blockIdx.x] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Centred Gram matrix  G = sum_i (x_i - c)(x_i - c)^T  for d = 64 / 128 on the fp64 matrix cores: the particle index is
// the contraction index (4 particles per MFMA), both operands come from one LDS tile of 32 centred rows, and only
// the block lower triangle is computed (wave w owns the row strips w and NB-1-w: NB + 1 blocks each, balanced).
// Row stride D + 16 doubles: the two rows a 32-lane group reads land on disjoint banks.
#define GRAM_TP 32

template <typename T, int D>
__global__ __launch_bounds__(64 * (D / 32)) void k_gram_mm(int64_t n, const T* __restrict__ x,
                                                          const double* __restrict__ center, double n_div,
                                                          double* __restrict__ partials) {
    extern __shared__ __align__(16) double tile[];
    constexpr int NB = D / 16, NT = 64 * (D / 32), STRIDE = D + 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ia = wave, ib = NB - 1 - wave;
    doublex4 accA[NB], accB[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) accA[j] = accB[j] = doublex4{0.0, 0.0, 0.0, 0.0};
    // the next tile's rows are fetched into registers while the current tile is in the matrix pipe
    constexpr int PER = GRAM_TP * D / NT;  // elements per thread and tile
    T nxt[PER];
    auto fetch = [&](int64_t t0) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int e = q * NT + threadIdx.x;
            const int r = e / D, c = e - r * D;
            nxt[q] = (t0 + r < n) ? x[(t0 + r) * D + c] : (T)0;
        }
    };
    // n_div > 0: `center` holds column SUMS and the centre is sum / n_div (k_center_from_sum's division, done here: the
    // reference fit needs no launch between the sums' all-reduce and this kernel)
    double cen[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int e = q * NT + threadIdx.x;
        const double cv = center[e % D];
        cen[q] = n_div > 0.0 ? cv / n_div : cv;
    }
    const int64_t tstride = (int64_t)gridDim.x * GRAM_TP;
    fetch((int64_t)blockIdx.x * GRAM_TP);
    for (int64_t t0 = (int64_t)blockIdx.x * GRAM_TP; t0 < n; t0 += tstride) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int e = q * NT + threadIdx.x;
            const int r = e / D, c = e - r * D;
            tile[r * STRIDE + c] = (t0 + r < n) ? (double)nxt[q] - cen[q] : 0.0;
        }
        __syncthreads();
        if (t0 + tstride < n) fetch(t0 + tstride);
#pragma unroll
        for (int t = 0; t < GRAM_TP / 4; t++) {
            const double* base = tile + (4 * t + (lane >> 4)) * STRIDE + (lane & 15);
            const double aA = base[16 * ia], aB = base[16 * ib];
#pragma unroll
            for (int jb = 0; jb < NB; jb++) {
                if (jb <= ib) {  // wave-uniform
                    const double b = base[16 * jb];
                    accB[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB, b, accB[jb], 0, 0, 0);
                    if (jb <= ia) accA[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA, b, accA[jb], 0, 0, 0);
                }
            }
        }
    }
    // block (i, j): lane holds G[16 i + (lane >> 4) + 4 r][16 j + (lane & 15)]; mirrored into the upper triangle
    double* out = partials + (size_t)blockIdx.x * D * D;
#pragma unroll
    for (int jb = 0; jb < NB; jb++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int col = 16 * jb + (lane & 15);
            if (jb <= ib) {
                const int row = 16 * ib + (lane >> 4) + 4 * r;
                out[(size_t)row * D + col] = accB[jb][r];
                if (jb != ib) out[(size_t)col * D + row] = accB[jb][r];
            }
            if (jb <= ia && ia != ib) {
                const int row = 16 * ia + (lane >> 4) + 4 * r;
                out[(size_t)row * D + col] = accA[jb][r];
                if (jb != ia) out[(size_t)col * D + row] = accA[jb][r];
            }
        }
    }
}

// d = 32 / 64 without the LDS tile: a lane loads its matrix-core operand element straight from the row-major rows - lane (k, i) of a
// group of four particles reads x[p + k][16 b + i] (sixteen lanes = 128 contiguous bytes, four such runs per load) for every
// coordinate block b, and since the A operand of block row I and the B operand of block column J have the same lane layout the
// same registers feed all lower blocks (I, J).  Every byte is loaded once and nothing is staged: 103 -> 60 us at 1M x 32 (the LDS
// version moved 2.4 TB/s), 516 -> 1xx us at 1M x 64.  Same particle-to-block assignment (32-row tiles) and the same accumulation
// order per accumulator as k_gram_mm<T, D>: the same bits.
template <typename T, int D>
__global__ __launch_bounds__(64) void k_gram_stream(int64_t n, const T* __restrict__ x, const double* __restrict__ center,
                                                   double n_div, double* __restrict__ partials) {
    constexpr int NB = D / 16, U = D == 32 ? 8 : 4, TRIPS = 8 / U;  // U groups of four particles per fetch; 32 rows per tile
    constexpr int NACC = NB * (NB + 1) / 2;
    const int lane = threadIdx.x, kq = lane >> 4, ci = lane & 15;
    double cen[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) cen[b] = n_div > 0.0 ? center[16 * b + ci] / n_div : center[16 * b + ci];  // (see k_gram_mm)
    doublex4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = doublex4{0.0, 0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * 32;
    T v[NB][U];
    auto fetch = [&](int64_t p0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t p = p0 + 4 * u + kq;
            const bool ok = p < n;
#pragma unroll
            for (int b = 0; b < NB; b++) v[b][u] = ok ? x[p * D + 16 * b + ci] : (T)0;
        }
    };
    fetch((int64_t)blockIdx.x * 32);
    for (int64_t t0 = (int64_t)blockIdx.x * 32; t0 < n; t0 += stride) {
#pragma unroll
        for (int h = 0; h < TRIPS; h++) {
            const int64_t p0 = t0 + 4 * U * h;
            double dv[NB][U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool ok = p0 + 4 * u + kq < n;
#pragma unroll
                for (int b = 0; b < NB; b++) dv[b][u] = ok ? (double)v[b][u] - cen[b] : 0.0;
            }
            // the next fetch is in flight while this one is in the matrix pipe
            const int64_t pn = h + 1 < TRIPS ? p0 + 4 * U : t0 + stride;
            if (pn < n) fetch(pn);
#pragma unroll
            for (int u = 0; u < U; u++) {
                // k_gram_mm's order within a group of four particles: for jb: accB[jb] (row strip NB - 1 - w) then accA[jb] - per
                // accumulator only the order over the groups matters, which is u ascending here as there
#pragma unroll
                for (int ib = 0; ib < NB; ib++)
#pragma unroll
                    for (int jb = 0; jb <= ib; jb++)
                        acc[ib * (ib + 1) / 2 + jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[ib][u], dv[jb][u], acc[ib * (ib + 1) / 2 + jb], 0, 0, 0);
            }
        }
    }
    // block (i, j): lane holds G[16 i + (lane >> 4) + 4 r][16 j + (lane & 15)]; mirrored into the upper triangle
    double* out = partials + (size_t)blockIdx.x * D * D;
#pragma unroll
    for (int ib = 0; ib < NB; ib++)
#pragma unroll
        for (int jb = 0; jb <= ib; jb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * ib + kq + 4 * r, col = 16 * jb + ci;
                const double val = acc[ib * (ib + 1) / 2 + jb][r];
                out[(size_t)row * D + col] = val;
                if (jb != ib) out[(size_t)col * D + row] = val;
            }
}

// fixed-order sum of the per-block partial matrices, one thread per matrix entry (coalesced across entries)
// sums the per-block partial matrices in a fixed order: 16 columns per block, 64 groups of partials per column in flight
__global__ __launch_bounds__(1024) void k_gram_mm_reduce(int nblocks, int ncols, const double* __restrict__ partials,
                                                         double* __restrict__ out, double* __restrict__ out2) {
    __shared__ double s[64][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    double v0 = 0.0, v1 = 0.0;
    if (col < ncols) {
        int b = g;
        for (; b + 64 < nblocks; b += 128) {
            v0 += partials[(size_t)b * ncols + col];
            v1 += partials[(size_t)(b + 64) * ncols + col];
        }
        if (b < nblocks) v0 += partials[(size_t)b * ncols + col];
    }
    s[g][c] = v0 + v1;
    __syncthreads();
    if (g == 0 && col < ncols) {
        double t = 0.0;
        for (int q = 0; q < 64; q++) t += s[q][c];
        out[col] = t;
        if (out2) out2[col] = t;  // (the copy asmc_reference_factor reads, ctx->d_ref + 128)
    }
}

bool asmc_gram_mm_supported(int d, const void* x) {
    static const bool no32 = getenv("ASMC_GRAM_RB32") != nullptr;
    return (d == 64 || d == 128 || (d == 32 && !no32)) && ((uintptr_t)x % 16 == 0);
}

// enqueues the Gram kernel and the reduction of its per-block partial matrices; the d x d result lands in ctx->d_partials
int asmc_gram_mm_launch(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* d_center, int* grid_out,
                        hipStream_t st, double* out2, double n_div) {
    int grid = (int)((n + GRAM_TP - 1) / GRAM_TP);
    // d = 32: one wave per block, so eight blocks per CU are needed to keep enough loads in flight
    static const int per_cu32 = getenv("ASMC_GRAM32_PER_CU") ? atoi(getenv("ASMC_GRAM32_PER_CU")) : 8;
    int cap = (d == 32 || (d == 64 && !getenv("ASMC_GRAM_LDS32"))) ? per_cu32 * ctx->num_cu : 2 * ctx->num_cu;  // (one wave per block: eight blocks per CU)  // (more blocks change nothing at d = 64 / 128: measured)
    if ((size_t)cap * d * d > ctx->gram_cap) cap = (int)(ctx->gram_cap / ((size_t)d * d));
    if (grid > cap) grid = cap;
    *grid_out = grid;
    const size_t lds = (size_t)GRAM_TP * (d + 16) * sizeof(double);
#define GRAM_CASE(TT, DD) \
    ASMC_LAUNCH(ctx, st, "k_gram_mm", (k_gram_mm<TT, DD>), dim3(grid), dim3(64 * (DD / 32)), lds, st, n, (const TT*)x, d_center, n_div, ctx->d_gram)
    static const bool lds_tile = getenv("ASMC_GRAM_LDS32") != nullptr;  // (the LDS-tile kernel at d = 32 / 64, for comparison)
    if ((d == 32 || d == 64) && !lds_tile) {
        if (x_dtype == ASMC_F64 && d == 32)
            ASMC_LAUNCH(ctx, st, "k_gram_mm", (k_gram_stream<double, 32>), dim3(grid), dim3(64), 0, st, n, (const double*)x, d_center, n_div, ctx->d_gram);
        else if (x_dtype == ASMC_F64)
            ASMC_LAUNCH(ctx, st, "k_gram_mm", (k_gram_stream<double, 64>), dim3(grid), dim3(64), 0, st, n, (const double*)x, d_center, n_div, ctx->d_gram);
        else if (d == 32)
            ASMC_LAUNCH(ctx, st, "k_gram_mm", (k_gram_stream<float, 32>), dim3(grid), dim3(64), 0, st, n, (const float*)x, d_center, n_div, ctx->d_gram);
        else
            ASMC_LAUNCH(ctx, st, "k_gram_mm", (k_gram_stream<float, 64>), dim3(grid), dim3(64), 0, st, n, (const float*)x, d_center, n_div, ctx->d_gram);
    } else if (x_dtype == ASMC_F64) {
        if (d == 128) GRAM_CASE(double, 128);
        else if (d == 64) GRAM_CASE(double, 64);
        else GRAM_CASE(double, 32);
    } else {
        if (d == 128) GRAM_CASE(float, 128);
        else if (d == 64) GRAM_CASE(float, 64);
        else GRAM_CASE(float, 32);
    }
#undef GRAM_CASE
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_gram_mm_reduce", k_gram_mm_reduce, dim3((d * d + 15) / 16), dim3(1024), 0, st, grid, d * d,
                (const double*)ctx->d_gram, ctx->d_partials, out2);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename T, int D, int NOISE, int MODE>
static int launch_mm(asmc_ctx* ctx, int64_t n, T* x, double* ll, double* lp, double* lq, const double* pack, const PcnDev& pd,
                     const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out, hipStream_t st) {
    constexpr bool XP = MODE == MM_XPROPOSE || MODE == MM_XPROPOSE_T;
    const size_t lds = XP ? ((size_t)2 * mm_ksum(D / 16) * 64 + D) * sizeof(double)
                     : MODE == MM_UNWHITEN_X ? ((size_t)mm_ksum(D / 16) * 64 + D) * sizeof(double)
                          : ((size_t)mm_ksum(D / 16) * 64 + D + (size_t)(pd.ll.C + pd.lp.C + pd.lq.C) * D * 2) * sizeof(double);
    ASMC_REQUIRE(lds <= 160 * 1024 - 512 - BM_TAB_N * sizeof(bm_d2), "operand image and density tables exceed the LDS");  // (512: the static part - block counts, log-weights)
    auto kern = k_pcn_mm<T, D, NOISE, MODE>;
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int per_cu = lds > 80 * 1024 ? 1 : 2;
    const int64_t n_groups = (n + 15) / 16;
    const int64_t want = (n_groups + MM_WAVES - 1) / MM_WAVES;
    int grid = (int)(want < (int64_t)ctx->num_cu * per_cu ? want : (int64_t)ctx->num_cu * per_cu);
    if (grid > ASMC_MAX_BLOCKS) grid = ASMC_MAX_BLOCKS;
    *grid_out = grid;
    ASMC_LAUNCH(ctx, st, XP ? "k_pcn_mm_propose" : MODE == MM_STEP ? "k_pcn_mm_step" : MODE == MM_STEP_T ? "k_tpcn_mm_step" : MODE == MM_WHITEN ? "k_pcn_mm_whiten" : "k_pcn_mm_unwhiten", kern,
                dim3(grid), dim3(MM_THREADS), lds, st, n, x, ll, lp, lq, pack, pd, rho_ptr, step, block_counts);
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
    if (mode == MM_XPROPOSE) { MM_CASE(TT, DD, ASMC_NOISE_F64, MM_XPROPOSE) }    \
    if (mode == MM_XPROPOSE_T) { MM_CASE(TT, DD, ASMC_NOISE_F64, MM_XPROPOSE_T) } \
    if (mode == MM_UNWHITEN) { MM_CASE(TT, DD, ASMC_NOISE_F64, MM_UNWHITEN) }    \
    if (mode == MM_UNWHITEN_X) { MM_CASE(TT, DD, ASMC_NOISE_F64, MM_UNWHITEN_X) } \
    if (mode == MM_STEP_T) {                                                     \
        if (pd.noise == ASMC_NOISE_F32) { MM_CASE(TT, DD, ASMC_NOISE_F32, MM_STEP_T) } \
        MM_CASE(TT, DD, ASMC_NOISE_F64, MM_STEP_T)                               \
    }                                                                            \
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
