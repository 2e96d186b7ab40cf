// asmc_pcn.hip — proposal draw, built-in densities, population moments, pCN mutation.
//
// Replaces (reference mj-will/aspire):
//   src/aspire/samplers/smc/minipcn.py:69-135  MiniPCNSMC.mutate  (wraps third-party minipcn —
//       absent from the reference tree; the kernel below implements THIS repository's pCN
//       specification; parity with minipcn is unpinned)
//   src/aspire/samplers/smc/base.py:507-519 + src/aspire/samples.py:1217-1219  tempered target
//       log p_t = (1-beta) log_q + beta (log_likelihood + log_prior) [+ log|J|], NaN -> -inf
//   src/aspire/samplers/mcmc.py:66-67  flow.sample_and_log_prob for the analytic Gaussian proposal
//
// Data movement: particle rows (d*s bytes, particle-major) are fetched with fully coalesced
// 16-B-per-lane loads into a padded LDS tile (64 rows per wave, row stride = rowbytes + 16 so that
// per-lane row reads are bank-conflict free), processed one particle per lane, and written back
// coalesced from the same tile.  Noise is generated in-kernel (Philox4x32-10 + Box-Muller), so one
// pCN step moves 2*d*s + 48 bytes per particle and nothing else.
#include <stdlib.h>

#include <chrono>

#include "asmc_common.h"
#include "asmc_tile.h"
#include "asmc_pcn_dev.h"
#include "asmc_transform_dev.h"

// =============================================================================================
// LDS row tiles
// =============================================================================================
// diagonal-mixture log-density of the row stored (as T) at `row`
template <typename T>
__device__ __forceinline__ double mixture_eval(const MixDev& m, int d, const char* row) {
    double best = -INFINITY;
    double terms[ASMC_MAX_COMPONENTS];
    const int C = m.C;
    for (int c = 0; c < C; c++) {
        double q = 0.0;
        const double* mu = m.mu + (size_t)c * d;
        const double* pr = m.prec + (size_t)c * d;
        for (int j = 0; j < d; j++) {
            const double t = row_get<T>(row, j) - mu[j];
            q = fma(t * t, pr[j], q);
        }
        const double v = m.logw[c] - 0.5 * q;
        terms[c] = v;
        best = fmax(best, v);
    }
    if (C == 1) return terms[0];
    if (best == -INFINITY) return -INFINITY;
    double s = 0.0;
    for (int c = 0; c < C; c++) s += exp(terms[c] - best);
    return best + log(s);
}

// =============================================================================================
// fused pCN step, generic in d: per-lane work vector v[k] lives in LDS (SoA: v[k*64 + lane])
// =============================================================================================

// PHASE 0: fused (propose + built-in targets + accept, in place)
// PHASE 1: propose only (writes x_prop tile, qform_old, qform_new)
template <typename T, int VEC, int PHASE>
__global__ __launch_bounds__(ASMC_BLOCK) void k_pcn_step(
    int64_t n, T* __restrict__ x, double* __restrict__ ll, double* __restrict__ lp, double* __restrict__ lq,
    PcnDev p, const double* __restrict__ rho_ptr, uint32_t step, long long* __restrict__ block_counts,
    T* __restrict__ x_prop, double* __restrict__ qf_old, double* __restrict__ qf_new, int waves_per_block) {
    extern __shared__ __align__(16) char smem[];
    const int d = p.d;
    const int rowbytes = d * (int)sizeof(T);
    const int ldsrow = lds_row_stride(rowbytes);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_bytes = 64 * ldsrow + 64 * 8 * d;
    char* tile = smem + (size_t)wave * wave_bytes;
    double* v = reinterpret_cast<double*>(tile + 64 * ldsrow) + lane;  // v[k*64]
    char* myrow = tile + lane * ldsrow;
    const double rho = *rho_ptr;
    const double a = sqrt(1.0 - rho * rho);
    long long n_acc = 0;
    const int64_t n_tiles = (n + 63) / 64;
    bm_d2* bmt = bm_lds();  // the Box-Muller tables (the first tile's __syncthreads publishes them)
    bm_tab_stage_rt(bmt, p.bmtab);
    for (int64_t tile0 = (int64_t)blockIdx.x * waves_per_block; tile0 < n_tiles;
         tile0 += (int64_t)gridDim.x * waves_per_block) {
        const int64_t t = tile0 + wave;
        const bool active = t < n_tiles;
        const int64_t i = t * 64 + lane;
        const bool valid = active && i < n;
        const int64_t row0 = t * 64;
        const int64_t valid_bytes = active ? (((n - row0) < 64 ? (n - row0) : 64) * (int64_t)rowbytes) : 0;
        if (active) tile_load<VEC>(reinterpret_cast<const char*>(x) + row0 * rowbytes, valid_bytes, rowbytes, ldsrow, tile, lane);
        __syncthreads();
        bool acc = false;
        if (valid) {
            const unsigned long long gid = p.gid0 + (unsigned long long)i;
            // dx -> v
            for (int j = 0; j < d; j++) v[j * 64] = row_get<T>(myrow, j) - p.mu[j];
            // y = Linv dx (in place, descending rows), q0 = |y|^2
            double q0 = 0.0;
            for (int j = d - 1; j >= 0; j--) {
                const double* Lr = p.Linv + (size_t)j * d;
                double s = 0.0;
                for (int k = 0; k <= j; k++) s = fma(Lr[k], v[k * 64], s);
                v[j * 64] = s;
                q0 = fma(s, s, q0);
            }
            // y' = a y + rho sqrt(s) xi (s = 1 for the Gaussian reference), q1 = |y'|^2
            double q1 = 0.0;
            const double rs = tpcn_scale(rho, p.nu, q0, p.gam, i);
            for (int qd = 0; 4 * qd < d; qd++) {
                double z[4];
                normal_quad(p.seed, gid, step, (uint32_t)qd, bmt, z[0], z[1], z[2], z[3]);
                for (int e = 0; e < 4 && 4 * qd + e < d; e++) {
                    const double ye = fma(rs, z[e], a * v[(4 * qd + e) * 64]);
                    v[(4 * qd + e) * 64] = ye;
                    q1 = fma(ye, ye, q1);
                }
            }
            // x' = mu + L y' (in place, descending rows); rounded to the storage type
            for (int j = d - 1; j >= 0; j--) {
                const double* Lr = p.L + (size_t)j * d;
                double s = 0.0;
                for (int k = 0; k <= j; k++) s = fma(Lr[k], v[k * 64], s);
                v[j * 64] = (double)(T)(p.mu[j] + s);
            }
            if (PHASE == 1) {
                // the accept kernel adds half of these to the tempered log-targets: twice the reference's correction
                qf_old[i] = 2.0 * ref_corr(q0, p.nu, d);
                qf_new[i] = 2.0 * ref_corr(q1, p.nu, d);
                for (int j = 0; j < d; j++) row_set<T>(myrow, j, v[j * 64]);
            } else {
                // the old row is still in `myrow`; stage x' into it only on acceptance, so evaluate
                // the targets from a scratch copy: write x' to the row AFTER saving nothing — the old
                // row is not needed again (old ll/lp/lq come from HBM), so overwrite and restore on reject
                const double oll = ll[i], olp = lp[i], olq = lq[i];
                // evaluate the built-in targets at x' directly from the work vector via a temp row view
                double nll, nlp, nlq;
                {
                    // temporarily keep the old row in registers-free form: swap through LDS element-wise
                    // (row <- x', v <- old x) so mixture_eval can read a contiguous typed row
                    for (int j = 0; j < d; j++) {
                        const double oldx = row_get<T>(myrow, j);
                        row_set<T>(myrow, j, v[j * 64]);
                        v[j * 64] = oldx;
                    }
                    nll = mixture_eval<T>(p.ll, d, myrow);
                    nlp = mixture_eval<T>(p.lp, d, myrow);
                    nlq = mixture_eval<T>(p.lq, d, myrow);
                }
                const double lpn = log_p_t(nll, nlp, nlq, p.beta);
                const double lpo = log_p_t(oll, olp, olq, p.beta);
                const double log_a = (lpn + ref_corr(q1, p.nu, d)) - (lpo + ref_corr(q0, p.nu, d));
                const double u = accept_uniform(p.seed, gid, step);
                acc = log(u) < log_a;
                if (acc) {
                    ll[i] = nll;
                    lp[i] = nlp;
                    lq[i] = nlq;
                    n_acc++;
                } else {
                    for (int j = 0; j < d; j++) row_set<T>(myrow, j, v[j * 64]);  // restore old row
                }
            }
        }
        __syncthreads();
        if (active) {
            if (PHASE == 1) {
                tile_store<VEC>(reinterpret_cast<char*>(x_prop) + row0 * rowbytes, valid_bytes, rowbytes, ldsrow, tile, lane);
            } else {
                const unsigned long long accmask = __ballot(acc);
                if (accmask != 0ULL)
                    tile_store_rows<VEC>(reinterpret_cast<char*>(x) + row0 * rowbytes, valid_bytes, rowbytes, ldsrow, tile, lane, accmask);
            }
        }
        __syncthreads();
    }
    if (PHASE == 0) {
        __shared__ long long s_cnt[ASMC_BLOCK / 64];
        n_acc = wave_sum_ll(n_acc);
        if (lane == 0) s_cnt[wave] = n_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tsum = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); w++) tsum += s_cnt[w];
            block_counts[blockIdx.x] = tsum;
        }
    }
}

// =============================================================================================
// fused pCN step, register resident (d == D, compile time): one particle per lane, the working
// vector v[D] lives in VGPRs, L / Linv / mu / target parameters are wave-uniform scalar loads, both
// triangular mat-vecs are fully unrolled FMA chains.  Rows travel HBM <-> LDS tile <-> registers
// with 16-B accesses on every hop.
// =============================================================================================
template <typename T, int D>
__device__ __forceinline__ void row_to_regs(const char* row, double (&v)[D]) {
    constexpr int ROWB = D * (int)sizeof(T);
#pragma unroll
    for (int c = 0; c < ROWB / 16; c++) {
        const uint4 q = *reinterpret_cast<const uint4*>(row + 16 * c);
        if (sizeof(T) == 8) {
            v[2 * c] = __longlong_as_double(((long long)q.y << 32) | (long long)q.x);
            v[2 * c + 1] = __longlong_as_double(((long long)q.w << 32) | (long long)q.z);
        } else {
            v[4 * c] = (double)__uint_as_float(q.x);
            v[4 * c + 1] = (double)__uint_as_float(q.y);
            v[4 * c + 2] = (double)__uint_as_float(q.z);
            v[4 * c + 3] = (double)__uint_as_float(q.w);
        }
    }
}

template <typename T, int D>
__device__ __forceinline__ void regs_to_row(char* row, const double (&v)[D]) {
    constexpr int ROWB = D * (int)sizeof(T);
#pragma unroll
    for (int c = 0; c < ROWB / 16; c++) {
        uint4 q;
        if (sizeof(T) == 8) {
            const long long a = __double_as_longlong(v[2 * c]), b = __double_as_longlong(v[2 * c + 1]);
            q.x = (uint32_t)a;
            q.y = (uint32_t)(a >> 32);
            q.z = (uint32_t)b;
            q.w = (uint32_t)(b >> 32);
        } else {
            q.x = __float_as_uint((float)v[4 * c]);
            q.y = __float_as_uint((float)v[4 * c + 1]);
            q.z = __float_as_uint((float)v[4 * c + 2]);
            q.w = __float_as_uint((float)v[4 * c + 3]);
        }
        *reinterpret_cast<uint4*>(row + 16 * c) = q;
    }
}

template <int D>
__device__ __forceinline__ double mixture_eval_regs(const MixDev& m, const double (&v)[D]) {
    // component 0 (the only one for plain Gaussians): no exp/log
    double q = 0.0;
#pragma unroll
    for (int j = 0; j < D; j++) {
        const double t = v[j] - m.mu[j];
        q = fma(t * t, m.prec[j], q);
    }
    double best = m.logw[0] - 0.5 * q;
    if (m.C == 1) return best;
    // further components: running (max, sum) log-sum-exp; runtime loop keeps the code small
    double s = 1.0;
#pragma unroll 1
    for (int c = 1; c < m.C; c++) {
        const double* __restrict__ mu = m.mu + c * D;
        const double* __restrict__ pr = m.prec + c * D;
        double qc = 0.0;
#pragma unroll
        for (int j = 0; j < D; j++) {
            const double t = v[j] - mu[j];
            qc = fma(t * t, pr[j], qc);
        }
        const double tc = m.logw[c] - 0.5 * qc;
        if (tc > best) {
            s = fma(s, exp(best - tc), 1.0);
            best = tc;
        } else if (tc > -INFINITY) {
            s += exp(tc - best);
        }
    }
    return best == -INFINITY ? -INFINITY : best + log(s);
}

// All read-only tables are packed into ONE `const __restrict__` buffer (ctx scratch, filled per mutate call):
// only a noalias kernel argument lets LLVM prove the tables are not clobbered by the particle stores and
// select scalar (s_load) instead of vector loads, and one base pointer keeps the SGPR budget for data.
// L and Linv are stored as PACKED lower triangles (row j at j(j+1)/2): 2 x 4.2 KB at d = 32, so that everything a
// step touches (~10 KB) stays inside the 16 KB scalar data cache (dense 2 x 8 KB tables thrashed it).
// layout (doubles): Ltri[D(D+1)/2] | Linvtri[D(D+1)/2] | mu[D] | 3 x { logw[8] | mu[8*D] | prec[8*D] }  (ll, lp, lq)
// v <- A v for a lower-triangular A stored packed by rows (wave-uniform, read through scalar loads), in place.
// Descending row groups of RG: rows j0-RG+1..j0 only read v[0..j0], which later (lower) groups never need
// overwritten entries of, so the update is legal in place.
template <int D>
__device__ __forceinline__ void tri_matvec_inplace(const double* __restrict__ A, double (&v)[D]) {
    constexpr int RG = D >= 4 ? 4 : D;
#pragma unroll
    for (int g = D / RG - 1; g >= 0; g--) {
        const int j0 = g * RG;  // rows j0 .. j0+RG-1
        double s[RG];
#pragma unroll
        for (int r = 0; r < RG; r++) s[r] = 0.0;
#pragma unroll
        for (int k = 0; k < j0 + RG; k++) {
#pragma unroll
            for (int r = 0; r < RG; r++)
                if (k <= j0 + r) s[r] = fma(A[(j0 + r) * (j0 + r + 1) / 2 + k], v[k], s[r]);
        }
#pragma unroll
        for (int r = 0; r < RG; r++) v[j0 + r] = s[r];
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    // each wave owns its LDS tile: ordering within the wave is enough (no s_barrier, waves run decoupled)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// MODE 0: one pCN step on x (whiten, propose, un-whiten, evaluate, accept)            — any built-in target
// MODE 1: one pCN step on the WHITENED state y = Linv (x - mu) carried in the x buffer: y' = a y + rho xi needs
//         no mat-vec, x' = mu + L y' is formed four rows at a time and consumed on the fly by the three
//         (single-component) quadratic forms, so it is never materialised: one mat-vec per step instead of two
// MODE 2: x -> y in place (start of a mutation);  MODE 3: y -> x in place + re-evaluate ll/lp/lq at the stored x
#define PCN_X_STEP 0
#define PCN_Y_STEP 1
#define PCN_WHITEN 2
#define PCN_UNWHITEN 3
#define PCN_UNWHITEN_X 4  // y -> x in place, carried ll/lp/lq untouched (flow-proposal path)
#define PCN_X_STEP_T 5    // PCN_X_STEP / PCN_Y_STEP with the Student-t reference (tpCN), selected at compile time
#define PCN_Y_STEP_T 6
// whitened state kept COORDINATE-MAJOR in a scratch buffer (ys[j * n_pad + i]) for the length of a mutation: the step
// kernel then needs no LDS staging at all (lane i reads 8 B of 64 consecutive particles per coordinate: 512-B coalesced
// loads straight into registers), so its occupancy is set by registers (3 waves/SIMD) instead of the 17 KB LDS tile
// (2 waves/SIMD), and rejected proposals cost no store.  x itself is only read by the first and written by the last.
#define PCN_WHITEN_S 7      // x (row-major) -> ys
#define PCN_Y_STEP_S 8      // step on ys
#define PCN_Y_STEP_TS 9     // ... Student-t reference
#define PCN_UNWHITEN_S 10   // ys -> x (row-major), ll / lp / lq re-evaluated at the stored x
#define PCN_UNWHITEN_XS 11  // ys -> x (row-major), carried ll / lp / lq untouched (flow-proposal path)
// proposal half of the split path (arbitrary Python callables evaluate the densities between propose and accept):
// x' -> the row-major buffer in p.ys, twice the reference's correction at y and y' -> the ll / lp arguments
#define PCN_X_PROPOSE 12
#define PCN_X_PROPOSE_T 13
// the same for 16 < d < 32 on the D = 32 kernel: rows of d elements, coordinates d..31 held at zero, tables padded with
// the identity (the generic LDS kernel needs 0.5-1.1 ms per proposal there, this one 0.3 ms)
#define PCN_X_PROPOSE_PAD 14
#define PCN_X_PROPOSE_PAD_T 15
// coordinate-major whitened-state step for densities with several mixture components: x' = mu + L y' is materialised in
// a second register array and handed to the general mixture evaluation (PCN_Y_STEP_S folds it into three quadratic
// forms on the fly, which only works for single Gaussians); still one mat-vec per step and no LDS
#define PCN_Y_STEP_SG 16
#define PCN_Y_STEP_TSG 17

// 64-row tile copies for rows of `dr` < D elements (PCN_X_PROPOSE_PAD): fully unrolled, element-wise coalesced
// accesses, row index by multiplication with magic = floor(2^32 / dr) + 1 (exact for e < 2^16) - no run-time loop, no
// division (a loop here brings back the scalar-load hoisting described above)
template <typename T, int D>
__device__ __forceinline__ void pad_tile_load(const T* __restrict__ g, int64_t valid_elems, int dr, unsigned magic, int ldsrow,
                                              char* lds, int lane) {
#pragma unroll
    for (int it = 0; it < D; it++) {
        const int e = it * 64 + lane;
        if (e < 64 * dr) {
            const int r = (int)__umulhi((unsigned)e, magic), c = e - r * dr;
            T v = (T)0;
            if (e < valid_elems) v = g[e];
            reinterpret_cast<T*>(lds + r * ldsrow)[c] = v;
        }
    }
}
template <typename T, int D>
__device__ __forceinline__ void pad_tile_store_rows(T* __restrict__ g, int64_t valid_elems, int dr, unsigned magic, int ldsrow,
                                                    const char* lds, int lane, unsigned long long rowmask) {
#pragma unroll
    for (int it = 0; it < D; it++) {
        const int e = it * 64 + lane;
        if (e < 64 * dr && e < valid_elems) {
            const int r = (int)__umulhi((unsigned)e, magic), c = e - r * dr;
            if ((rowmask >> r) & 1ULL) g[e] = reinterpret_cast<const T*>(lds + r * ldsrow)[c];
        }
    }
}

template <typename T, int D, int NOISE, int MODE>
__device__ __forceinline__ void pcn_reg_body(int64_t n, T* __restrict__ x, double* __restrict__ ll,
                                             double* __restrict__ lp, double* __restrict__ lq,
                                             const double* __restrict__ ptab, const PcnScalars& p,
                                             const double* __restrict__ rho_ptr, uint32_t step,
                                             long long* __restrict__ block_counts) {
    extern __shared__ __align__(16) char smem[];
    constexpr bool TP = MODE == PCN_X_STEP_T || MODE == PCN_Y_STEP_T || MODE == PCN_Y_STEP_TS || MODE == PCN_X_PROPOSE_T ||
                        MODE == PCN_X_PROPOSE_PAD_T || MODE == PCN_Y_STEP_TSG;
    constexpr bool GEN = MODE == PCN_Y_STEP_SG || MODE == PCN_Y_STEP_TSG;  // general mixtures on the whitened state
    constexpr bool SOA = (MODE >= PCN_WHITEN_S && MODE <= PCN_UNWHITEN_XS) || GEN;
    constexpr bool PAD = MODE == PCN_X_PROPOSE_PAD || MODE == PCN_X_PROPOSE_PAD_T;
    constexpr bool PROPOSE = MODE == PCN_X_PROPOSE || MODE == PCN_X_PROPOSE_T || PAD;
    constexpr int M = PROPOSE ? PCN_X_STEP : GEN ? PCN_Y_STEP : MODE == PCN_WHITEN_S ? PCN_WHITEN : (MODE == PCN_Y_STEP_S || MODE == PCN_Y_STEP_TS) ? PCN_Y_STEP
                      : MODE == PCN_UNWHITEN_S ? PCN_UNWHITEN : MODE == PCN_UNWHITEN_XS ? PCN_UNWHITEN_X
                      : TP ? MODE - PCN_X_STEP_T : MODE;
    constexpr bool ROW_IN = !SOA || M == PCN_WHITEN;     // the state arrives as row-major x through the LDS tile
    constexpr bool ROW_OUT = !SOA || M == PCN_UNWHITEN || M == PCN_UNWHITEN_X;  // ... leaves that way
    T* __restrict__ ys = reinterpret_cast<T*>(p.ys);
    constexpr int ROWB = D * (int)sizeof(T);
    constexpr int LDSROW = ROWB + 16;
    const int WPB = (int)(blockDim.x >> 6);  // waves per block: chosen by the launcher from the LDS budget
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dr = PAD ? p.d_real : D;                                       // elements per stored row
    const int dn = (p.d_noise > 0 && p.d_noise < D) ? p.d_noise : D;         // real dimension of a zero-padded problem (asmc_pcn_mutate)
    const int rowb = PAD ? dr * (int)sizeof(T) : ROWB;
    const int ldsrow = PAD ? lds_row_stride(rowb) : LDSROW;
    char* tile = smem + (size_t)wave * 64 * LDSROW;
    char* myrow = tile + lane * ldsrow;
    const double rho = *rho_ptr;
    const double a = sqrt(1.0 - rho * rho);
    long long n_acc = 0;
    const int64_t n_tiles = (n + 63) / 64;
    // the Box-Muller tables of the default noise, staged by the whole block (the only block-wide barrier of the kernel)
    bm_d2* bmt = nullptr;
    if constexpr (NOISE == ASMC_NOISE_F64 && (M == PCN_X_STEP || M == PCN_Y_STEP)) {
        bmt = bm_lds();
        bm_tab_stage_rt(bmt, p.bmtab);
        __syncthreads();
    }
    // one 64-particle tile per wave and NO tile loop: with a loop LLVM hoists the ~1300 loop-invariant
    // table loads and the fp64 polynomial constants out of it and then spills them
    const int64_t t = (int64_t)blockIdx.x * WPB + wave;
    if (t < n_tiles) {
        const bool active = true;
        const int64_t row0 = t * 64;
        const int64_t i = row0 + lane;
        // coordinate-major state: wave-uniform tile base (scalar registers) + the lane as a 32-bit offset
        // descriptor over the whole scratch buffer, built from wave-uniform values only (< 4 GB: checked by the host)
        const unsigned long long ysa = (unsigned long long)(uintptr_t)ys;
        T* ysu = reinterpret_cast<T*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ysa >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ysa));
        // after fused flow steps a tile's state lives in the half of the allocation its parity byte names (asmc_pcn_fused.hip)
        if (M == PCN_UNWHITEN_X && SOA && p.tile_par != nullptr && __builtin_amdgcn_readfirstlane((int)p.tile_par[t]) != 0)
            ysu += (size_t)__builtin_amdgcn_readfirstlane((int)(unsigned)p.n_pad) * D;
        const __amdgpu_buffer_rsrc_t ysr = __builtin_amdgcn_make_buffer_rsrc(
            ysu, 0, SOA ? (int)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)p.n_pad * D * sizeof(T))) : 0,
            0x00020000);
        const unsigned ys_tile = (unsigned)__builtin_amdgcn_readfirstlane((int)t) * 64u * (unsigned)sizeof(T);
        const unsigned ys_row = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p.n_pad) * (unsigned)sizeof(T);
        const unsigned ys_lane = (unsigned)lane * (unsigned)sizeof(T);
        const bool valid = active && i < n;
        const int64_t valid_bytes = active ? (((n - row0) < 64 ? (n - row0) : 64) * (int64_t)rowb) : 0;
        char* gbase = reinterpret_cast<char*>(x) + row0 * rowb;
        const unsigned magic = PAD ? 0xFFFFFFFFu / (unsigned)dr + 1u : 0u;
        if (PAD)
            pad_tile_load<T, D>(reinterpret_cast<const T*>(gbase), valid_bytes / (int64_t)sizeof(T), dr, magic, ldsrow, tile, lane);
        else if (active && ROW_IN)
            tile_load<16>(gbase, valid_bytes, ROWB, LDSROW, tile, lane);
        double oll = 0.0, olp = 0.0, olq = 0.0;
        if (valid && !PROPOSE && (M == PCN_X_STEP || M == PCN_Y_STEP)) {
            oll = ll[i];
            olp = lp[i];
            olq = lq[i];
        }
        wave_lds_sync();
        bool acc = false;
        const long zoff = 0;
        const double* __restrict__ tab = ptab + zoff;
        const double* __restrict__ Lp = tab;
        const double* __restrict__ Lip = tab + PTAB_TRI(D);
        const double* __restrict__ mup = tab + 2 * PTAB_TRI(D);
        const double* __restrict__ m0 = tab + 2 * PTAB_TRI(D) + D;
        const MixDev mll = {p.c_ll, m0, m0 + ASMC_MAX_COMPONENTS, m0 + ASMC_MAX_COMPONENTS * (1 + D)};
        const MixDev mlp = {p.c_lp, m0 + PTAB_MIX(D), m0 + PTAB_MIX(D) + ASMC_MAX_COMPONENTS,
                            m0 + PTAB_MIX(D) + ASMC_MAX_COMPONENTS * (1 + D)};
        const MixDev mlq = {p.c_lq, m0 + 2 * PTAB_MIX(D), m0 + 2 * PTAB_MIX(D) + ASMC_MAX_COMPONENTS,
                            m0 + 2 * PTAB_MIX(D) + ASMC_MAX_COMPONENTS * (1 + D)};
        if (valid) {
            const unsigned long long gid = p.gid0 + (unsigned long long)i;
            double v[D];
            if (PAD) {
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = j < dr ? row_get<T>(myrow, j) : 0.0;  // padded mu is zero too
            } else if (ROW_IN) {
                row_to_regs<T, D>(myrow, v);
            } else {
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = soa_load<T>(ysr, ys_lane, ys_tile + (unsigned)j * ys_row);
            }
            if (M == PCN_WHITEN) {
#pragma unroll
                for (int j = 0; j < D; j++) v[j] -= mup[j];
                tri_matvec_inplace<D>(Lip, v);
                if (SOA) {
#pragma unroll
                    for (int j = 0; j < D; j++) soa_store<T>(ysr, ys_lane, ys_tile + (unsigned)j * ys_row, v[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < D; j++) v[j] = (double)(T)v[j];
                    regs_to_row<T, D>(myrow, v);
                    acc = true;
                }
            } else if (M == PCN_UNWHITEN || M == PCN_UNWHITEN_X) {
                tri_matvec_inplace<D>(Lp, v);
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = (double)(T)(mup[j] + v[j]);
                if (M == PCN_UNWHITEN) {
                    ll[i] = mixture_eval_regs<D>(mll, v);
                    lp[i] = mixture_eval_regs<D>(mlp, v);
                    lq[i] = mixture_eval_regs<D>(mlq, v);
                }
                regs_to_row<T, D>(myrow, v);
                acc = true;
            } else {
                double q0 = 0.0;
                if (M == PCN_X_STEP) {
#pragma unroll
                    for (int j = 0; j < D; j++) v[j] -= mup[j];
                    // y = Linv (x - mu), in place, 4 interleaved FMA chains per row group
#ifndef ASMC_ABLATE_MATVEC
                    tri_matvec_inplace<D>(Lip, v);
#endif
                }
#pragma unroll
                for (int j = 0; j < D; j++) q0 = fma(v[j], v[j], q0);
                // y' = a y + rho sqrt(s) xi (rounded to the storage type when y is what gets stored); q1 = |y'|^2
                double q1 = 0.0;
                const double rs = tpcn_scale_ct<TP>(rho, p.nu, q0, p.gam, i);
                if (NOISE == ASMC_NOISE_F64) {
#pragma unroll
                    for (int qd = 0; qd < (D + 3) / 4; qd++) {
                        double z[4];
                        normal_quad(p.seed, gid, step, (uint32_t)qd, bmt, z[0], z[1], z[2], z[3]);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (4 * qd + e < D) {
                                v[4 * qd + e] = fma(rs, z[e], a * v[4 * qd + e]);
                                if (M == PCN_Y_STEP) v[4 * qd + e] = (double)(T)v[4 * qd + e];
                                q1 = fma(v[4 * qd + e], v[4 * qd + e], q1);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);  // one block at a time (register pressure)
                    }
                } else {
#pragma unroll
                    for (int qd = 0; qd < D / 4; qd++) {
                        double z[4];
#ifndef ASMC_ABLATE_NOISE
                        normal_quad_f32(p.seed, gid, step, (uint32_t)qd, z[0], z[1], z[2], z[3]);
#else
                        z[0] = z[1] = z[2] = z[3] = 0.25 * (double)(lane + qd);
#endif
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            v[4 * qd + e] = fma(rs, z[e], a * v[4 * qd + e]);
                            if (M == PCN_Y_STEP) v[4 * qd + e] = (double)(T)v[4 * qd + e];
                            q1 = fma(v[4 * qd + e], v[4 * qd + e], q1);
                        }
                        if (qd & 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (PAD || dn < D) {  // the padded coordinates carry no noise: y' = 0 there, and they stay out of |y'|^2
                    const int dm = PAD ? dr : dn;  // (dn < D: a wave-uniform branch that only zero-padded problems take)
                    q1 = 0.0;
#pragma unroll
                    for (int j = 0; j < D; j++) {
                        v[j] = j < dm ? v[j] : 0.0;
                        q1 = fma(v[j], v[j], q1);
                    }
                }
                double nll, nlp, nlq;
                if (M == PCN_X_STEP) {
                    // x' = mu + L y', rounded to the storage type
#ifndef ASMC_ABLATE_MATVEC
                    tri_matvec_inplace<D>(Lp, v);
#endif
#pragma unroll
                    for (int j = 0; j < D; j++) v[j] = (double)(T)(mup[j] + v[j]);
                    if (PROPOSE) {
                        ll[i] = 2.0 * ref_corr_ct<TP>(q0, p.nu, PAD ? dr : dn);
                        lp[i] = 2.0 * ref_corr_ct<TP>(q1, p.nu, PAD ? dr : dn);
                        if (PAD) {
#pragma unroll
                            for (int j = 0; j < D; j++)
                                if (j < dr) row_set<T>(myrow, j, v[j]);
                        } else {
                            regs_to_row<T, D>(myrow, v);
                        }
                        acc = true;
                    }
#ifndef ASMC_ABLATE_TARGET
                    if (!PROPOSE) {
                        nll = mixture_eval_regs<D>(mll, v);
                        nlp = mixture_eval_regs<D>(mlp, v);
                        nlq = mixture_eval_regs<D>(mlq, v);
                    } else {
                        nll = nlp = nlq = 0.0;
                    }
#else
                    nll = v[0], nlp = v[1], nlq = v[2];
#endif
                } else if (GEN) {
                    // v keeps y'; x' goes to a second register array for the general mixture evaluation
                    double xq[D];
#pragma unroll
                    for (int g = 0; g < D / 4; g++) {
                        const int j0 = 4 * g;
                        double sr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int k = 0; k < j0 + 4; k++) {
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (k <= j0 + r) sr[r] = fma(Lp[(j0 + r) * (j0 + r + 1) / 2 + k], v[k], sr[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++) xq[j0 + r] = (double)(T)(mup[j0 + r] + sr[r]);
                    }
                    nll = mixture_eval_regs<D>(mll, xq);
                    nlp = mixture_eval_regs<D>(mlp, xq);
                    nlq = mixture_eval_regs<D>(mlq, xq);
                } else {
                    // v keeps y' (it is what gets stored); x'_j = mu_j + sum_k L[j,k] y'_k is produced four rows
                    // at a time and folded straight into the three quadratic forms (component 0 of each target)
                    double qa = 0.0, qb = 0.0, qc = 0.0;
                    // L comes from the BLOCKED copy of the table (PTAB_BLK): sixteen consecutive doubles per 4 x 4 block, in the order of
                    // use.  Read from the row-packed triangle the scalar loads were merged into 64-byte pieces that straddle row groups -
                    // data loaded long before its use, which the pCN instantiation then parked in vector-register lanes: 1 134
                    // v_writelane / v_readlane per tile next to 1 092 fp64 FMAs (round 5's generated code).  Same products, same order.
                    const double* __restrict__ Lb = tab + PTAB_SIZE(D);
#pragma unroll
                    for (int g = 0; g < D / 4; g++) {
                        const int j0 = 4 * g;
                        double sr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int k = 0; k < j0 + 4; k++) {
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (k <= j0 + r) sr[r] = fma(Lb[16 * (g * (g + 1) / 2 + k / 4) + 4 * (k % 4) + r], v[k], sr[r]);
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int j = j0 + r;
                            const double xj = (double)(T)(mup[j] + sr[r]);
                            const double ta = xj - mll.mu[j], tb = xj - mlp.mu[j], tc = xj - mlq.mu[j];
                            qa = fma(ta * ta, mll.prec[j], qa);
                            qb = fma(tb * tb, mlp.prec[j], qb);
                            qc = fma(tc * tc, mlq.prec[j], qc);
                        }
                    }
                    nll = mll.logw[0] - 0.5 * qa;
                    nlp = mlp.logw[0] - 0.5 * qb;
                    nlq = mlq.logw[0] - 0.5 * qc;
                }
                const double lpn = log_p_t(nll, nlp, nlq, p.beta);
                const double lpo = log_p_t(oll, olp, olq, p.beta);
                const double log_a = (lpn + ref_corr_ct<TP>(q1, p.nu, dn)) - (lpo + ref_corr_ct<TP>(q0, p.nu, dn));
                const double u = accept_uniform(p.seed, gid, step);
                if (!PROPOSE) acc = log(u) < log_a;
                if (acc && !PROPOSE) {
                    if (ROW_OUT) {
                        regs_to_row<T, D>(myrow, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < D; j++) soa_store<T>(ysr, ys_lane, ys_tile + (unsigned)j * ys_row, v[j]);
                    }
                    ll[i] = nll;
                    lp[i] = nlp;
                    lq[i] = nlq;
                    n_acc++;
                }
            }
        }
        wave_lds_sync();
        {
            const unsigned long long accmask = ROW_OUT ? __ballot(acc) : 0ULL;
            char* obase = PROPOSE ? reinterpret_cast<char*>(p.ys) + row0 * rowb : gbase;
            if (PAD) {
                if (accmask != 0ULL)
                    pad_tile_store_rows<T, D>(reinterpret_cast<T*>(obase), valid_bytes / (int64_t)sizeof(T), dr, magic, ldsrow, tile,
                                              lane, accmask);
            } else if (accmask != 0ULL) {
                tile_store_rows<16>(obase, valid_bytes, ROWB, LDSROW, tile, lane, accmask);
            }
        }
        wave_lds_sync();
    }
    if (!PROPOSE && (M == PCN_X_STEP || M == PCN_Y_STEP)) {
        __shared__ long long s_cnt[ASMC_BLOCK / 64];
        n_acc = wave_sum_ll(n_acc);
        if (lane == 0) s_cnt[wave] = n_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tsum = 0;
            for (int w = 0; w < WPB; w++) tsum += s_cnt[w];
            block_counts[blockIdx.x] = tsum;
        }
    }
}

template <typename T, int D, int NOISE, int MODE>
__global__ __launch_bounds__(ASMC_BLOCK, 2) void k_pcn_reg(int64_t n, T* __restrict__ x, double* __restrict__ ll,
                                                          double* __restrict__ lp, double* __restrict__ lq,
                                                          const double* __restrict__ ptab, PcnScalars p,
                                                          const double* __restrict__ rho_ptr, uint32_t step,
                                                          long long* __restrict__ block_counts) {
    pcn_reg_body<T, D, NOISE, MODE>(n, x, ll, lp, lq, ptab, p, rho_ptr, step, block_counts);
}
// (a three-waves-per-SIMD build of the LDS-free step modes spills 200-300 B/lane to scratch and is slower: 0.19 vs 0.136 ms; round 6,
// with the blocked table: the pCN step at 168 registers spills 52 B and takes 165 us against 148 at two waves; the Student-t step
// fits three waves by itself - profiles/r06_ab_pcn_blocked_L.txt)

// Flow-proposal pCN on the whitened state, split around the flow's log-density kernel (asmc_flow.hip):
//   PCN_FLOW_PROPOSE  y' = a y + rho xi;  x' = mu + L y'  -> x_prop tile, ll'(x'), lp'(x') (built-in targets)
//   PCN_FLOW_ACCEPT   regenerates y' from the same Philox counters (no y' round trip through HBM), reads
//                     ll', lp', lq' and accepts: y <- y' (accepted rows only), carried log-probs updated
#define PCN_FLOW_PROPOSE 0
#define PCN_FLOW_ACCEPT 1
#define PCN_FLOW_PROPOSE_S 2  // the same with y coordinate-major in p.ys (see PCN_*_S above): propose stages only x'
#define PCN_FLOW_ACCEPT_S 3   // through LDS, accept touches no LDS at all
#define PCN_FLOW_PROPOSE_SX 4 // PCN_FLOW_PROPOSE_S without the built-in densities: the proposal half of the whitened-state
                              // split session, whose densities come from the caller (arbitrary Python callables)
#define PCN_FLOW_ACCEPT_SJ 5  // PCN_FLOW_ACCEPT_S with a carried log-Jacobian (chain in a preconditioned space): the two
                              // arrays arrive through the y / x_prop parameters, which coordinate-major accept does not use
// PCN_FLOW_PROPOSE_SX whose proposal lives in a preconditioned space z = T(x) (reference
// transforms.py:294-316 inside smc/minipcn.py:105-119): the inverse transform x' = T^-1(z') and its log-Jacobian are
// applied to the register-resident proposal, and - when the proposal flow's data transform shares T's bounded stage -
// log q(x') is evaluated from z' in the same pass (asmc_mixture_logpdf_premap's arithmetic).  x' goes to x_prop, log|J| to
// ll_new, log q to lp_new; lq_new carries the packed table (TRTAB_* below).  One variant per bounded stage.
#define PCN_FLOW_PROPOSE_SXT_LOGIT 6
#define PCN_FLOW_PROPOSE_SXT_PROBIT 7
// packed table of the transform epilogue: TRTAB_ROWS rows of D doubles, then TRTAB_SCALARS scalars
//   rows: kind, lower, upper, mean, std, 1/(upper-lower), 1/std, premap a, b, lo, hi, h, q mean, q precision
//   scalars: eps, unit_logj, affine_logj, has_affine, q log-weight, has_q, minus_logj
#define TRTAB_ROWS 14
#define TRTAB_SCALARS 8

template <typename T, int D, int NOISE, int MODE>
__global__ __launch_bounds__(ASMC_BLOCK, 2) void k_pcn_reg_flow(
    int64_t n, T* __restrict__ y, T* __restrict__ x_prop, double* __restrict__ ll, double* __restrict__ lp,
    double* __restrict__ lq, double* __restrict__ ll_new, double* __restrict__ lp_new, const double* __restrict__ lq_new,
    const double* __restrict__ ptab, PcnScalars p, const double* __restrict__ rho_ptr, uint32_t step,
    long long* __restrict__ block_counts) {
    extern __shared__ __align__(16) char smem[];
    constexpr bool SOA = MODE >= PCN_FLOW_PROPOSE_S;
    constexpr bool TR = MODE == PCN_FLOW_PROPOSE_SXT_LOGIT || MODE == PCN_FLOW_PROPOSE_SXT_PROBIT;
    constexpr bool NO_DENS = MODE == PCN_FLOW_PROPOSE_SX || TR;
    constexpr bool HAS_LJ = MODE == PCN_FLOW_ACCEPT_SJ;
    constexpr int M = NO_DENS ? PCN_FLOW_PROPOSE : HAS_LJ ? PCN_FLOW_ACCEPT : SOA ? MODE - PCN_FLOW_PROPOSE_S : MODE;
    constexpr int ROWB = D * (int)sizeof(T);
    constexpr int LDSROW = ROWB + 16;
    const int WPB = (int)(blockDim.x >> 6);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tile = smem + (size_t)wave * 64 * LDSROW;
    char* myrow = tile + lane * LDSROW;
    const double rho = *rho_ptr;
    const double a = sqrt(1.0 - rho * rho);
    long long n_acc = 0;
    const int64_t n_tiles = (n + 63) / 64;
    bm_d2* bmt = nullptr;  // Box-Muller tables of the default noise (see pcn_reg_body)
    if constexpr (NOISE == ASMC_NOISE_F64) {
        bmt = bm_lds();
        bm_tab_stage_rt(bmt, p.bmtab);
        __syncthreads();
    }
    const int64_t t = (int64_t)blockIdx.x * WPB + wave;  // one tile per wave, no loop (see k_pcn_reg)
    if (t < n_tiles) {
        const int64_t row0 = t * 64;
        const int64_t i = row0 + lane;
        const bool valid = i < n;
        const int64_t valid_bytes = ((n - row0) < 64 ? (n - row0) : 64) * (int64_t)ROWB;
        if (!SOA) tile_load<16>(reinterpret_cast<const char*>(y) + row0 * ROWB, valid_bytes, ROWB, LDSROW, tile, lane);
        // coordinate-major state through one buffer descriptor (see soa_load / soa_store)
        const unsigned long long ysa = (unsigned long long)(uintptr_t)p.ys;
        T* ysu = reinterpret_cast<T*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ysa >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ysa));
        const __amdgpu_buffer_rsrc_t ysr = __builtin_amdgcn_make_buffer_rsrc(
            ysu, 0, SOA ? (int)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)p.n_pad * D * sizeof(T))) : 0,
            0x00020000);
        const unsigned ys_tile = (unsigned)__builtin_amdgcn_readfirstlane((int)t) * 64u * (unsigned)sizeof(T);
        const unsigned ys_row = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p.n_pad) * (unsigned)sizeof(T);
        const unsigned ys_lane = (unsigned)lane * (unsigned)sizeof(T);
        double oll = 0.0, olp = 0.0, olq = 0.0, nll = 0.0, nlp = 0.0, nlq = 0.0;
        double olj = 0.0, nlj = 0.0;
        double* const lj = HAS_LJ ? static_cast<double*>(static_cast<void*>(y)) : nullptr;
        if (valid && M == PCN_FLOW_ACCEPT) {
            oll = ll[i], olp = lp[i], olq = lq[i];
            nll = ll_new[i], nlp = lp_new[i], nlq = lq_new[i];
            if (HAS_LJ) olj = lj[i], nlj = static_cast<const double*>(static_cast<const void*>(x_prop))[i];
        }
        wave_lds_sync();
        bool acc = false;
        const double* __restrict__ Lp = ptab;
        const double* __restrict__ mup = ptab + 2 * PTAB_TRI(D);
        const double* __restrict__ m0 = ptab + 2 * PTAB_TRI(D) + D;
        const MixDev mll = {p.c_ll, m0, m0 + ASMC_MAX_COMPONENTS, m0 + ASMC_MAX_COMPONENTS * (1 + D)};
        const MixDev mlp = {p.c_lp, m0 + PTAB_MIX(D), m0 + PTAB_MIX(D) + ASMC_MAX_COMPONENTS,
                            m0 + PTAB_MIX(D) + ASMC_MAX_COMPONENTS * (1 + D)};
        if (valid) {
            const unsigned long long gid = p.gid0 + (unsigned long long)i;
            double v[D];
            if (SOA) {
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = soa_load<T>(ysr, ys_lane, ys_tile + (unsigned)j * ys_row);
            } else {
                row_to_regs<T, D>(myrow, v);
            }
            double q0 = 0.0, q1 = 0.0;
#pragma unroll
            for (int j = 0; j < D; j++) q0 = fma(v[j], v[j], q0);
            const double rs = tpcn_scale(rho, p.nu, q0, p.gam, i);  // the same variate in both modes
            if (NOISE == ASMC_NOISE_F64) {
#pragma unroll
                for (int qd = 0; qd < (D + 3) / 4; qd++) {
                    double z[4];
                    normal_quad(p.seed, gid, step, (uint32_t)qd, bmt, z[0], z[1], z[2], z[3]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (4 * qd + e < D) {
                            v[4 * qd + e] = (double)(T)fma(rs, z[e], a * v[4 * qd + e]);
                            q1 = fma(v[4 * qd + e], v[4 * qd + e], q1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int qd = 0; qd < D / 4; qd++) {
                    double z[4];
                    normal_quad_f32(p.seed, gid, step, (uint32_t)qd, z[0], z[1], z[2], z[3]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[4 * qd + e] = (double)(T)fma(rs, z[e], a * v[4 * qd + e]);
                        q1 = fma(v[4 * qd + e], v[4 * qd + e], q1);
                    }
                    if (qd & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (M == PCN_FLOW_PROPOSE) {
                tri_matvec_inplace<D>(Lp, v);
#pragma unroll
                for (int j = 0; j < D; j++) v[j] = (double)(T)(mup[j] + v[j]);
                if (!NO_DENS) {
                    ll_new[i] = mixture_eval_regs<D>(mll, v);
                    lp_new[i] = mixture_eval_regs<D>(mlp, v);
                }
                if (TR) {
                    constexpr int HINTS = ASMC_TR_NO_PERIODIC |
                                          (MODE == PCN_FLOW_PROPOSE_SXT_LOGIT ? ASMC_TR_NO_PROBIT : ASMC_TR_NO_LOGIT);
                    const double* __restrict__ tt = lq_new;
                    const double* __restrict__ sc = tt + TRTAB_ROWS * D;
                    const double eps = sc[0];
                    const bool has_affine = sc[3] != 0.0, has_q = sc[5] != 0.0;
                    double lj_b = 0.0, quad = 0.0, qq = 0.0;
                    bool any_b = false;
#pragma unroll
                    for (int j = 0; j < D; j++) {
                        if (has_q) {  // log q(x') from z' (asmc_mixture_logpdf_premap)
                            const double tq = clip(v[j] * tt[7 * D + j] + tt[8 * D + j], tt[9 * D + j], tt[10 * D + j]);
                            quad = fma(tt[11 * D + j] * tq, tq, quad);
                            const double dq = tq - tt[12 * D + j];
                            qq = fma(dq * dq, tt[13 * D + j], qq);
                        }
                        CoordPar c;
                        c.kind = (int)tt[j], c.periodic = 0;
                        c.lo = tt[D + j], c.up = tt[2 * D + j], c.mean = tt[3 * D + j], c.std = tt[4 * D + j];
                        c.inv_w = tt[5 * D + j], c.inv_std = tt[6 * D + j];
                        any_b |= c.kind != 0;
                        v[j] = (double)(T)transform_coord<1, HINTS>(v[j], c, has_affine, eps, lj_b);
                        // keep the table reads of later coordinates behind this one's arithmetic (hoisted, they cost 100 VGPRs)
                        if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    double lj = 0.0;  // k_transform's grouping: affine constant, then the bounded block
                    if (has_affine) lj += -sc[2];
                    if (any_b) lj += lj_b + (-sc[1]);
                    ll_new[i] = lj;
                    if (has_q) lp_new[i] = ((sc[4] - 0.5 * qq) + quad) - (sc[6] != 0.0 ? lj : 0.0);
                }
                regs_to_row<T, D>(myrow, v);
                acc = true;
            } else {
                double lpn = log_p_t(nll, nlp, nlq, p.beta);
                double lpo = log_p_t(oll, olp, olq, p.beta);
                if (HAS_LJ) {  // as k_pcn_accept_flags: the log-Jacobian joins the tempered log-target, NaN -> -inf
                    lpn += nlj, lpo += olj;
                    lpn = log_p_t_guard(lpn);
                    lpo = log_p_t_guard(lpo);
                }
                const double log_a = (lpn + ref_corr(q1, p.nu, D)) - (lpo + ref_corr(q0, p.nu, D));
                const double u = accept_uniform(p.seed, gid, step);
                acc = log(u) < log_a;
                if (acc) {
                    if (HAS_LJ) lj[i] = nlj;
                    if (SOA) {
#pragma unroll
                        for (int j = 0; j < D; j++) soa_store<T>(ysr, ys_lane, ys_tile + (unsigned)j * ys_row, v[j]);
                    } else {
                        regs_to_row<T, D>(myrow, v);
                    }
                    ll[i] = nll;
                    lp[i] = nlp;
                    lq[i] = nlq;
                    n_acc++;
                }
            }
        }
        wave_lds_sync();
        {
            const unsigned long long accmask = (SOA && M == PCN_FLOW_ACCEPT) ? 0ULL : __ballot(acc);
            char* obase = reinterpret_cast<char*>(M == PCN_FLOW_PROPOSE ? x_prop : y) + row0 * ROWB;
            if (accmask != 0ULL) tile_store_rows<16>(obase, valid_bytes, ROWB, LDSROW, tile, lane, accmask);
        }
        wave_lds_sync();
    }
    if (M == PCN_FLOW_ACCEPT) {
        __shared__ long long s_cnt[ASMC_BLOCK / 64];
        n_acc = wave_sum_ll(n_acc);
        if (lane == 0) s_cnt[wave] = n_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tsum = 0;
            for (int w = 0; w < WPB; w++) tsum += s_cnt[w];
            block_counts[blockIdx.x] = tsum;
        }
    }
}

// tpCN: unit-scale Gamma(shape) variates of every particle for the Markov steps step .. step + nsteps - 1 (Marsaglia-Tsang,
// counter based: a step's variates do not depend on the chain, so several steps are drawn by one launch)
template <bool F32>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gamma_draw(int64_t n, double shape, unsigned long long seed,
                                                          unsigned long long gid0, uint32_t step, int nsteps, int64_t stride_n,
                                                          double* __restrict__ out, const double* __restrict__ bmtab) {
    bm_d2* bmt = nullptr;
    if constexpr (!F32) {
        bmt = bm_lds();
        bm_tab_stage<ASMC_BLOCK>(bmt, bmtab);
        __syncthreads();
    }
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride)
        for (int s = 0; s < nsteps; s++)
            out[(size_t)s * stride_n + i] = gamma_unit<F32>(shape, seed, gid0 + (unsigned long long)i, step + (uint32_t)s, bmt);
}

// before a step kernel: points pd.gam at the step's scale variates in ctx->d_gamma (tpCN only), drawing the next
// ASMC_GAMMA_BATCH steps' worth when the step is not among those already there (one launch per eight steps instead of one
// per step: 15 us of every 150 us step)
static int pcn_prepare_gamma(asmc_ctx* ctx, int64_t n, PcnDev& pd, uint32_t step, hipStream_t st) {
    if (!(pd.nu > 0.0)) {
        pd.gam = nullptr;
        return ASMC_OK;
    }
    const double shape = 0.5 * ((double)(pd.d_noise > 0 ? pd.d_noise : pd.d) + pd.nu);
    const int64_t stride_n = (n + 63) / 64 * 64;
    const bool hit = ctx->gam_count > 0 && ctx->gam_n == n && ctx->gam_seed == (unsigned long long)pd.seed &&
                     ctx->gam_gid0 == (unsigned long long)pd.gid0 && ctx->gam_shape == shape && ctx->gam_noise == pd.noise &&
                     step >= ctx->gam_step0 && step - ctx->gam_step0 < (uint32_t)ctx->gam_count;
    if (!hit) {
        int batch = (int)((int64_t)ASMC_GAMMA_BATCH * ctx->n_max / stride_n);  // what the buffer holds
        if (batch > ASMC_GAMMA_BATCH) batch = ASMC_GAMMA_BATCH;
        if (batch < 1) batch = 1;
        static const bool single = getenv("ASMC_GAMMA_PER_STEP") != nullptr;
        if (single) batch = 1;
        const int grid = grid_for(n, ASMC_BLOCK, ASMC_MAX_BLOCKS * 4);
        if (pd.noise == ASMC_NOISE_F32)
            ASMC_LAUNCH(ctx, st, "k_gamma_draw", k_gamma_draw<true>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, shape,
                        (unsigned long long)pd.seed, (unsigned long long)pd.gid0, step, batch, stride_n, ctx->d_gamma,
                        (const double*)ctx->d_bmtab);
        else
            ASMC_LAUNCH(ctx, st, "k_gamma_draw", k_gamma_draw<false>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, shape,
                        (unsigned long long)pd.seed, (unsigned long long)pd.gid0, step, batch, stride_n, ctx->d_gamma,
                        (const double*)ctx->d_bmtab);
        ASMC_LAUNCH_CHECK();
        ctx->gam_n = n, ctx->gam_seed = (unsigned long long)pd.seed, ctx->gam_gid0 = (unsigned long long)pd.gid0;
        ctx->gam_shape = shape, ctx->gam_noise = pd.noise, ctx->gam_step0 = step, ctx->gam_count = batch;
    }
    pd.gam = ctx->d_gamma + (size_t)(step - ctx->gam_step0) * stride_n;
    return ASMC_OK;
}

// sums the per-block accept counts of step `t`, records them, adapts the step size
// (log rho += (acc - target)/(t+1)^0.75, rho clipped to [1e-4, 0.99])
// adapt: 0 = off, 1 = after every step, k >= 2 = LAGGED (sampler_kwargs["adapt_lag"] = k): the step size is held for blocks of k
// steps; the step that ends a block (t + 1 = 0 mod k, or `last`: the final step of the call) applies the block's updates in
// order, each with its own count and step index - the same arithmetic k steps late.  A sharded run then exchanges its accept
// counts once per block (k cells in one all-reduce) instead of once per step.
// cells != NULL (sharded, lagged): block_counts is unused; the GLOBAL counts of the block's steps sit in cells[t' % k] and are
// recorded here (counts_out, rho_hist) together with the replay.
__global__ __launch_bounds__(1024) void k_pcn_adapt(int nblocks, const long long* __restrict__ block_counts,
                                                   int64_t n, int t, long long* __restrict__ counts_out,
                                                   double* __restrict__ rho_ptr, double* __restrict__ rho_hist,
                                                   double target, int adapt, const double* __restrict__ rho_src, int last,
                                                   const long long* __restrict__ cells) {
    // rho_src: where the step size step t used is kept when it is not *rho_ptr (fused flow steps of sharded runs, which
    // adapt in the next step's prologue: *rho_ptr is only brought up to date here, at the end of a call)
    __shared__ long long s_c[16];
    long long c = 0;
    if (cells == nullptr)
        for (int b = threadIdx.x; b < nblocks; b += 1024) c += block_counts[b];
    c = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = 0;
        for (int w = 0; w < 16; w++) c += s_c[w];
        const double rho = rho_src ? *rho_src : *rho_ptr;
        if (cells == nullptr) {
            counts_out[t] = c;
            rho_hist[t] = rho;
        }
        if (adapt <= 1) {
            *rho_ptr = adapt ? pcn_adapt_rho(rho, c, n, target, t) : rho;
        } else if ((t + 1) % adapt == 0 || last) {
            double r = rho;
            for (int tp = t - (t % adapt); tp <= t; tp++) {
                const long long ct = cells ? cells[tp % adapt] : (tp == t ? c : counts_out[tp]);
                if (cells) counts_out[tp] = ct, rho_hist[tp] = rho;
                r = pcn_adapt_rho(r, ct, n, target, tp);
            }
            *rho_ptr = r;
        } else {
            *rho_ptr = rho;
        }
    }
}

__global__ void k_set_scalar(double* __restrict__ cell, double v) { *cell = v; }

__global__ __launch_bounds__(1024) void k_count_sum(int nblocks, const long long* __restrict__ block_counts,
                                                   long long* __restrict__ cell) {  // (the caller points `cell` at the step's own cell)
    __shared__ long long s_c[16];
    long long c = 0;
    for (int b = threadIdx.x; b < nblocks; b += 1024) c += block_counts[b];
    c = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        c = 0;
        for (int w = 0; w < 16; w++) c += s_c[w];
        cell[0] = c;
    }
}

// closes step t: accept count (summed across ranks through the exchange hook when one is installed), history, adaptation.
// `last`: t is the final step of the call (a lagged adaptation closes its open block there).
static int pcn_close_step(asmc_ctx* ctx, hipStream_t st, int grid, const long long* d_block, int64_t n, int t,
                          long long* d_counts, double* d_rho, double* d_rho_hist, double target, int adapt, int last = 1) {
    if (ctx->count_hook) {
        const int lag = adapt >= 2 ? adapt : 1;
        if (lag > ctx->count_cells) {
            asmc_set_error("adapt_lag %d needs %d exchange cells, the installed hook has %d (asmc_pcn_set_count_cells)", lag, lag, ctx->count_cells);
            return ASMC_ERR_ARG;
        }
        ASMC_LAUNCH(ctx, st, "k_count_sum", k_count_sum, dim3(1), dim3(1024), 0, st, grid, d_block, ctx->count_cell + (lag > 1 ? t % lag : 0));
        ASMC_LAUNCH_CHECK();
        if (lag > 1 && !((t + 1) % lag == 0 || last)) return ASMC_OK;  // the block's counts are exchanged together, at its end
        const int hrc = ctx->count_hook(ctx->count_hook_user, reinterpret_cast<asmc_stream>(st));
        if (hrc != 0) {
            asmc_set_error("accept-count exchange hook failed (%d)", hrc);
            return ASMC_ERR_ARG;
        }
        ASMC_LAUNCH(ctx, st, "k_pcn_adapt", k_pcn_adapt, dim3(1), dim3(1024), 0, st, 1, (const long long*)ctx->count_cell,
                    ctx->count_n_global, t, d_counts, d_rho, d_rho_hist, target, adapt, (const double*)nullptr, last,
                    lag > 1 ? (const long long*)ctx->count_cell : (const long long*)nullptr);
    } else {
        ASMC_LAUNCH(ctx, st, "k_pcn_adapt", k_pcn_adapt, dim3(1), dim3(1024), 0, st, grid, d_block, n, t, d_counts, d_rho,
                    d_rho_hist, target, adapt, (const double*)nullptr, last, (const long long*)nullptr);
    }
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// split-path accept: per particle decision + scalar update, then a flat conditional row copy
__global__ __launch_bounds__(ASMC_BLOCK) void k_pcn_accept_flags(
    int64_t n, double* __restrict__ ll, double* __restrict__ lp, double* __restrict__ lq,
    const double* __restrict__ ll_new, const double* __restrict__ lp_new, const double* __restrict__ lq_new,
    double* __restrict__ lj_old, const double* __restrict__ lj_new, const double* __restrict__ qf_old,
    const double* __restrict__ qf_new, double beta, unsigned long long seed, unsigned long long gid0,
    uint32_t step, unsigned char* __restrict__ flags, unsigned long long* __restrict__ count) {
    long long c = 0;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        double lpn = log_p_t(ll_new[i], lp_new[i], lq_new[i], beta);
        double lpo = log_p_t(ll[i], lp[i], lq[i], beta);
        if (lj_new) {
            lpn += lj_new[i];
            lpn = log_p_t_guard(lpn);
        }
        if (lj_old) {
            lpo += lj_old[i];
            lpo = log_p_t_guard(lpo);
        }
        const double log_a = (lpn + 0.5 * qf_new[i]) - (lpo + 0.5 * qf_old[i]);
        const double u = accept_uniform(seed, gid0 + (unsigned long long)i, step);
        const bool acc = log(u) < log_a;
        flags[i] = acc ? 1 : 0;
        if (acc) {
            ll[i] = ll_new[i];
            lp[i] = lp_new[i];
            lq[i] = lq_new[i];
            if (lj_old && lj_new) lj_old[i] = lj_new[i];  // the carried log-Jacobian follows the accepted state
            c++;
        }
    }
    // one atomic per block (thousands of same-address atomics serialise at the L2: 8192 per-wave adds cost ~80 us)
    __shared__ long long s_c[ASMC_BLOCK / 64];
    c = wave_sum_ll(c);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < ASMC_BLOCK / 64; w++) t += s_c[w];
        if (t) atomicAdd(count, (unsigned long long)t);
    }
}

template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_copy_flagged_rows(int64_t n, int d, T* __restrict__ x,
                                                                 const T* __restrict__ x_prop,
                                                                 const unsigned char* __restrict__ flags) {
    const int64_t total = n * d;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t e = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / d;
        if (flags[r]) x[e] = x_prop[e];
    }
}

// the same in 16-byte pieces, `cpr` pieces per row (row bytes a multiple of 16, n * cpr < 2^32): the row index comes from
// a multiplication with magic = floor(2^32 / cpr) + 1 instead of a 64-bit division per element
__global__ __launch_bounds__(ASMC_BLOCK) void k_copy_flagged_rows16(unsigned total, int shift, unsigned magic,
                                                                   uint4* __restrict__ x, const uint4* __restrict__ x_prop,
                                                                   const unsigned char* __restrict__ flags) {
    const unsigned stride = gridDim.x * ASMC_BLOCK;
    for (unsigned e = blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const unsigned r = shift >= 0 ? e >> shift : (unsigned)(((unsigned long long)e * magic) >> 32);
        if (flags[r]) x[e] = x_prop[e];
    }
}

static int launch_copy_flagged(asmc_ctx* ctx, int64_t n, int d, int x_dtype, void* x, const void* x_prop,
                               const unsigned char* flags, hipStream_t st) {
    const int64_t rowb = (int64_t)d * (x_dtype == ASMC_F64 ? 8 : 4);
    const int64_t cpr = rowb / 16;
    // power-of-two piece counts shift; otherwise e * (floor(2^32 / cpr) + 1) >> 32 == e / cpr, exact while e * cpr < 2^32
    const bool pow2 = cpr > 0 && (cpr & (cpr - 1)) == 0;
    int shift = -1;
    if (pow2)
        for (shift = 0; (1LL << shift) < cpr; shift++) {
        }
    if (rowb % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)x_prop % 16) == 0 && n * cpr < (1LL << 32) &&
        (pow2 || n * cpr * cpr < (1LL << 32))) {
        const unsigned total = (unsigned)(n * cpr);
        const int grid = grid_for(total, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 4);
        ASMC_LAUNCH(ctx, st, "k_copy_flagged_rows", k_copy_flagged_rows16, dim3(grid), dim3(ASMC_BLOCK), 0, st, total, shift,
                    (unsigned)(0xFFFFFFFFu / (unsigned)cpr + 1u), (uint4*)x, (const uint4*)x_prop, flags);
    } else {
        const int grid2 = grid_for(n * d, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
        if (x_dtype == ASMC_F64)
            ASMC_LAUNCH(ctx, st, "k_copy_flagged_rows<double>", k_copy_flagged_rows<double>, dim3(grid2), dim3(ASMC_BLOCK), 0, st, n, d,
                        (double*)x, (const double*)x_prop, flags);
        else
            ASMC_LAUNCH(ctx, st, "k_copy_flagged_rows<float>", k_copy_flagged_rows<float>, dim3(grid2), dim3(ASMC_BLOCK), 0, st, n, d,
                        (float*)x, (const float*)x_prop, flags);
    }
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

// =============================================================================================
// analytic Gaussian proposal draw, built-in density evaluation
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gaussian_draw(int64_t n, int d, const double* __restrict__ mu,
                                                             const double* __restrict__ sigma,
                                                             unsigned long long seed, unsigned long long gid0,
                                                             uint32_t draw_id, T* __restrict__ x,
                                                             const double* __restrict__ bmtab) {
    bm_d2* bmt = bm_lds();
    bm_tab_stage<ASMC_BLOCK>(bmt, bmtab);
    __syncthreads();
    const int quads = (d + 3) / 4;  // one Philox block = four coordinates (asmc_pcn_dev.h normal_quad)
    const int64_t total = n * quads;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t e = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / quads;
        const int qd = (int)(e - i * quads);
        double z[4];
        normal_quad(seed, gid0 + (unsigned long long)i, draw_id, (uint32_t)qd, bmt, z[0], z[1], z[2], z[3]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = 4 * qd + c;
            if (j < d) x[i * d + j] = (T)fma(sigma[j], z[c], mu[j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gaussian_logq(int64_t n, int d, const double* __restrict__ mu,
                                                             const double* __restrict__ sigma,
                                                             const T* __restrict__ x, double* __restrict__ lq) {
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; i < n; i += stride) {
        double q = 0.0, ls = 0.0;
        for (int j = 0; j < d; j++) {
            const double z = ((double)x[i * d + j] - mu[j]) / sigma[j];
            q = fma(z, z, q);
            ls += log(sigma[j]);
        }
        lq[i] = -0.5 * q - ls - 0.5 * (double)d * 1.8378770664093454835606594728112;  // log(2 pi)
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(ASMC_BLOCK) void k_mixture_logpdf(int64_t n, int d, const T* __restrict__ x,
                                                              MixDev m, double* __restrict__ out,
                                                              int waves_per_block) {
    extern __shared__ __align__(16) char smem[];
    const int rowbytes = d * (int)sizeof(T);
    const int ldsrow = lds_row_stride(rowbytes);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tile = smem + (size_t)wave * 64 * ldsrow;
    const int64_t n_tiles = (n + 63) / 64;
    for (int64_t tile0 = (int64_t)blockIdx.x * waves_per_block; tile0 < n_tiles;
         tile0 += (int64_t)gridDim.x * waves_per_block) {
        const int64_t t = tile0 + wave;
        const bool active = t < n_tiles;
        const int64_t i = t * 64 + lane;
        const int64_t row0 = t * 64;
        const int64_t valid_bytes = active ? (((n - row0) < 64 ? (n - row0) : 64) * (int64_t)rowbytes) : 0;
        if (active) tile_load<VEC>(reinterpret_cast<const char*>(x) + row0 * rowbytes, valid_bytes, rowbytes, ldsrow, tile, lane);
        __syncthreads();
        if (active && i < n) out[i] = mixture_eval<T>(m, d, tile + lane * ldsrow);
        __syncthreads();
    }
}

// Flat form for rows of a power-of-two number (<= 64) of 16-byte pieces: one piece per thread, coalesced 16-byte loads with
// no LDS, the piece's coordinates' (mu, prec) of every component in registers for the whole grid-stride loop, quadratic forms
// completed by a butterfly over the row's lanes, log-sum-exp over the components by the row's first lane (same formula as
// mixture_eval; the quadratic form is summed in butterfly order instead of coordinate order: ~1e-16 relative).
template <typename T, int CMAX>
__global__ __launch_bounds__(ASMC_BLOCK) void k_mixture_flat(int64_t n, int d, int tpr_log2, const uint4* __restrict__ x, MixDev m,
                                                            double* __restrict__ out, const double* __restrict__ premap) {
    // premap != NULL (asmc_mixture_logpdf_premap): the density is evaluated at t_j = clip(a_j x_j + b_j, lo_j, hi_j) and
    // sum_j h_j t_j^2 is added; rows a, b, lo, hi, h of d doubles each
    constexpr int EPT = 16 / (int)sizeof(T);
    const int tpr = 1 << tpr_log2, C = m.C;
    const int64_t total = n << tpr_log2;
    const int64_t stride = (int64_t)gridDim.x * ASMC_BLOCK;  // a multiple of tpr: a thread keeps its coordinates
    const int c0 = (int)(((int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x) & (tpr - 1));
    double mu[CMAX][EPT], pr[CMAX][EPT];
#pragma unroll
    for (int c = 0; c < CMAX; c++)
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            mu[c][k] = c < C ? m.mu[(size_t)c * d + c0 * EPT + k] : 0.0;
            pr[c][k] = c < C ? m.prec[(size_t)c * d + c0 * EPT + k] : 0.0;
        }
    double pa[EPT], pb[EPT], plo[EPT], phi[EPT], ph[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int j = c0 * EPT + k;
        pa[k] = premap ? premap[j] : 1.0;
        pb[k] = premap ? premap[d + j] : 0.0;
        plo[k] = premap ? premap[2 * d + j] : -INFINITY;
        phi[k] = premap ? premap[3 * d + j] : INFINITY;
        ph[k] = premap ? premap[4 * d + j] : 0.0;
    }
    for (int64_t e0 = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e0 - (threadIdx.x & 63) < total; e0 += stride) {
        const bool valid = e0 < total;
        double q[CMAX], extra = 0.0;
#pragma unroll
        for (int c = 0; c < CMAX; c++) q[c] = 0.0;
        if (valid) {
            const uint4 raw = x[e0];
            const T* vals = reinterpret_cast<const T*>(&raw);
            double xv[EPT];
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                xv[k] = (double)vals[k];
                if (premap) {
                    xv[k] = fmin(fmax(xv[k] * pa[k] + pb[k], plo[k]), phi[k]);
                    extra = fma(ph[k] * xv[k], xv[k], extra);
                }
            }
#pragma unroll
            for (int c = 0; c < CMAX; c++)
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    const double t = xv[k] - mu[c][k];
                    q[c] = fma(t * t, pr[c][k], q[c]);
                }
        }
#pragma unroll
        for (int c = 0; c < CMAX; c++)
            for (int o = tpr >> 1; o >= 1; o >>= 1) q[c] += __shfl_xor(q[c], o, 64);
        if (premap)
            for (int o = tpr >> 1; o >= 1; o >>= 1) extra += __shfl_xor(extra, o, 64);
        if (valid && c0 == 0) {
            double best = -INFINITY, terms[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; c++) {
                terms[c] = c < C ? m.logw[c] - 0.5 * q[c] : -INFINITY;
                best = fmax(best, terms[c]);
            }
            double r = terms[0];
            if (C > 1) {
                if (best == -INFINITY) {
                    r = -INFINITY;
                } else {
                    double ssum = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; c++)
                        if (c < C) ssum += exp(terms[c] - best);
                    r = best + log(ssum);
                }
            }
            out[e0 >> tpr_log2] = r + extra;
        }
    }
}


// =============================================================================================
// population moments
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_colsum(int64_t n, int d, const T* __restrict__ x,
                                                      double* __restrict__ partials) {
    // thread t owns column (t % d) when ASMC_BLOCK % d == 0, otherwise a strided element walk with
    // per-element column lookup; partial sums are combined through LDS atomics-free reduction.
    extern __shared__ __align__(16) char smem[];
    double* s_acc = reinterpret_cast<double*>(smem);  // [ASMC_BLOCK]
    const int rows_per_pass = ASMC_BLOCK / d;         // >= 1 (d <= 256)
    const int my_col = threadIdx.x % d;
    const int my_sub = threadIdx.x / d;
    double acc = 0.0;
    if (my_sub < rows_per_pass) {
        for (int64_t r = (int64_t)blockIdx.x * rows_per_pass + my_sub; r < n; r += (int64_t)gridDim.x * rows_per_pass)
            acc += (double)x[r * d + my_col];
    }
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < d) {
        double v = 0.0;
        for (int s = 0; s < rows_per_pass; s++) v += s_acc[s * d + threadIdx.x];
        partials[(size_t)blockIdx.x * d + threadIdx.x] = v;
    }
}

// gram partial: G[j,k] += (x_ij - c_j)(x_ik - c_k) over the block's rows; tile of 64 centred rows in LDS
template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gram(int64_t n, int d, const T* __restrict__ x,
                                                    const double* __restrict__ center,
                                                    double* __restrict__ partials) {
    extern __shared__ __align__(16) char smem[];
    double* s_rows = reinterpret_cast<double*>(smem);  // [64][d+1]
    const int ld = d + 1;
    const int dd = d * d;
    // entries e = tid, tid+256, ... (< d*d <= 4096): at most 16 accumulators per thread
    double acc[16];
    int jj[16], kk[16];
#pragma unroll
    for (int a = 0; a < 16; a++) {
        acc[a] = 0.0;
        const int e = threadIdx.x + a * ASMC_BLOCK;
        const int ec = e < dd ? e : 0;
        jj[a] = ec / d;
        kk[a] = ec - jj[a] * d;
    }
    for (int64_t row0 = (int64_t)blockIdx.x * 64; row0 < n; row0 += (int64_t)gridDim.x * 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * d; e += ASMC_BLOCK) {
            const int r = e / d, c = e - r * d;
            const int64_t gr = row0 + r;
            s_rows[r * ld + c] = gr < n ? (double)x[gr * d + c] - center[c] : 0.0;
        }
        __syncthreads();
#pragma unroll 2
        for (int r = 0; r < 64; r++) {
            const double* row = s_rows + r * ld;
#pragma unroll
            for (int a = 0; a < 16; a++)
                if (a * ASMC_BLOCK < dd) acc[a] = fma(row[jj[a]], row[kk[a]], acc[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 16; a++) {
        const int e = threadIdx.x + a * ASMC_BLOCK;
        if (e < dd) partials[(size_t)blockIdx.x * dd + e] = acc[a];
    }
}

// Register-blocked centred Gram matrix: one wave owns a (8*BLK) x (8*BLK) quadrant of G, lane (bi, bj) a
// BLK x BLK block of it (16 or 64 accumulators in VGPRs).  Rows are staged 64 at a time through the wave's
// LDS tile with coalesced 16-B loads; per row every lane reads two BLK-wide slices (same row for all lanes:
// LDS broadcast, conflict free) and issues BLK^2 FMAs — 0.5 (BLK=4) / 0.25 (BLK=8) LDS reads per FMA.
// blockIdx.y selects the quadrant (d > 8*BLK needs several).  Block partial [d_pad x d_pad] per block.
template <typename T, int BLK>
__global__ __launch_bounds__(ASMC_BLOCK) void k_gram_rb(int64_t n, int d, const T* __restrict__ x,
                                                       const double* __restrict__ center,
                                                       double* __restrict__ partials, int n_quad_side) {
    extern __shared__ __align__(16) char smem[];
    constexpr int Q = 8 * BLK;  // quadrant side
    const int rowbytes = d * (int)sizeof(T);
    const int ldsrow = lds_row_stride(rowbytes);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* tile = smem + (size_t)wave * 64 * ldsrow;
    const int qi = blockIdx.y / n_quad_side, qj = blockIdx.y % n_quad_side;
    const int bi = lane >> 3, bj = lane & 7;
    const int i0 = qi * Q + bi * BLK, j0 = qj * Q + bj * BLK;  // first row / column of this lane's block
    double ci[BLK], cj[BLK], acc[BLK][BLK];
#pragma unroll
    for (int a = 0; a < BLK; a++) {
        ci[a] = (i0 + a < d) ? center[i0 + a] : 0.0;
        cj[a] = (j0 + a < d) ? center[j0 + a] : 0.0;
#pragma unroll
        for (int b = 0; b < BLK; b++) acc[a][b] = 0.0;
    }
    const int64_t n_tiles = (n + 63) / 64;
    const int wpb = (int)(blockDim.x >> 6);
    for (int64_t t = (int64_t)blockIdx.x * wpb + wave; t < n_tiles; t += (int64_t)gridDim.x * wpb) {
        const int64_t row0 = t * 64;
        const int rows = (int)((n - row0) < 64 ? (n - row0) : 64);
        wave_lds_sync();
        tile_load<16>(reinterpret_cast<const char*>(x) + row0 * rowbytes, (int64_t)rows * rowbytes, rowbytes, ldsrow, tile, lane);
        wave_lds_sync();
        for (int r = 0; r < rows; r++) {
            const T* row = reinterpret_cast<const T*>(tile + r * ldsrow);
            double ai[BLK], aj[BLK];
#pragma unroll
            for (int a = 0; a < BLK; a++) {
                ai[a] = (i0 + a < d) ? (double)row[i0 + a] - ci[a] : 0.0;
                aj[a] = (j0 + a < d) ? (double)row[j0 + a] - cj[a] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < BLK; a++)
#pragma unroll
                for (int b = 0; b < BLK; b++) acc[a][b] = fma(ai[a], aj[b], acc[a][b]);
        }
    }
    // combine the block's waves through LDS (fixed order), write the block partial of this quadrant
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);  // [wpb][Q*Q] fits: Q*Q*8 <= 64*ldsrow for d >= Q/2
    double* mine = red + (size_t)wave * Q * Q;
#pragma unroll
    for (int a = 0; a < BLK; a++)
#pragma unroll
        for (int b = 0; b < BLK; b++) mine[(bi * BLK + a) * Q + bj * BLK + b] = acc[a][b];
    __syncthreads();
    const int dpad = n_quad_side * Q;
    for (int e = threadIdx.x; e < Q * Q; e += (int)blockDim.x) {
        double v = red[e];
        for (int w = 1; w < wpb; w++) v += red[(size_t)w * Q * Q + e];
        const int gi = qi * Q + e / Q, gj = qj * Q + e % Q;
        partials[(size_t)blockIdx.x * dpad * dpad + (size_t)gi * dpad + gj] = v;
    }
}

// one block per column; 64 threads (one wave: the order every caller has always had) or 256 (the gather's column-sum partials:
// 4 096 rows - 24 us with one wave): strided partial sums, the wave's butterfly, then the waves in order
// keep / center (optional): the sum also goes to keep[col] (the copy asmc_reference_factor reads in ctx->d_ref) and
// center[col] = sum / n_mean (k_center_from_sum's division)
__global__ __launch_bounds__(256) void k_reduce_columns(int nblocks, int ncols, const double* __restrict__ partials,
                                                       double* __restrict__ out, double* __restrict__ keep = nullptr,
                                                       double* __restrict__ center = nullptr, double n_mean = 1.0) {
    __shared__ double s_w[4];
    const int nt = (int)blockDim.x;
    for (int col = blockIdx.x; col < ncols; col += gridDim.x) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += nt) v += partials[(size_t)b * ncols + col];
        v = wave_sum(v);
        if (nt != 64) {
            if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
            __syncthreads();
            v = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[col] = v;
            if (keep) keep[col] = v;
            if (center) center[col] = v / n_mean;
        }
    }
}

// =============================================================================================
// host side
// =============================================================================================
static int pick_vec(int rowbytes, const void* p0, const void* p1) {
    const uintptr_t a = (uintptr_t)p0 | (uintptr_t)p1;
    if (rowbytes % 16 == 0 && a % 16 == 0) return 16;
    if (rowbytes % 8 == 0 && a % 8 == 0) return 8;
    return 4;
}

static int waves_for_lds(size_t per_wave_bytes, size_t* lds_bytes_out) {
    const size_t budget = 160 * 1024 - 1024 - BM_TAB_N * 16;  // (the block's Box-Muller tables sit next to the tiles)
    int w = ASMC_BLOCK / 64;
    while (w > 1 && per_wave_bytes * (size_t)w > budget) w >>= 1;
    *lds_bytes_out = per_wave_bytes * (size_t)w;
    return w;
}

// gather the caller's tables into the ctx parameter block (one tiny kernel, stream ordered)
__global__ __launch_bounds__(256) void k_pcn_pack(PcnDev pd, double* __restrict__ t, double* __restrict__ rho_cell, double rho) {
    if (rho_cell && threadIdx.x == 0) *rho_cell = rho;  // (the call's starting step size rides along: no k_set_scalar launch)
    const int D = pd.dpad > pd.d ? pd.dpad : pd.d, dr = pd.d;  // tables of the D-dimensional kernel, identity beyond dr
    const int tri = D * (D + 1) / 2;
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int j = e / D, k = e - j * D;
        if (k <= j) {
            const bool in = j < dr;  // (k <= j < dr)
            t[j * (j + 1) / 2 + k] = in ? pd.L[j * dr + k] : (k == j ? 1.0 : 0.0);
            t[tri + j * (j + 1) / 2 + k] = in ? pd.Linv[j * dr + k] : (k == j ? 1.0 : 0.0);
        }
    }
    for (int e = threadIdx.x; e < D; e += 256) t[2 * tri + e] = e < dr ? pd.mu[e] : 0.0;
    if (D % 4 == 0) {  // the blocked copy of L (PTAB_BLK)
        double* tb = t + 2 * tri + D + 3 * PTAB_MIX(D);
        const int nblk = (D / 4) * (D / 4 + 1) / 2;
        for (int e = threadIdx.x; e < nblk * 16; e += 256) {
            const int b = e >> 4, kk = (e >> 2) & 3, r = e & 3;
            int g = 0;
            while ((g + 1) * (g + 2) / 2 <= b) g++;
            const int c = b - g * (g + 1) / 2, j = 4 * g + r, k = 4 * c + kk;
            tb[e] = k > j ? 0.0 : (j < dr ? pd.L[j * dr + k] : (k == j ? 1.0 : 0.0));
        }
    }
    double* m0 = t + 2 * tri + D;
    const MixDev* mixes[3] = {&pd.ll, &pd.lp, &pd.lq};
    for (int i = 0; i < 3; i++) {
        double* b = m0 + (size_t)i * PTAB_MIX(D);
        const int C = mixes[i]->C;
        for (int e = threadIdx.x; e < C; e += 256) b[e] = mixes[i]->logw[e];
        for (int e = threadIdx.x; e < C * D; e += 256) {
            b[ASMC_MAX_COMPONENTS + e] = mixes[i]->mu[e];
            b[ASMC_MAX_COMPONENTS * (1 + D) + e] = mixes[i]->prec[e];
        }
    }
}

static int pack_pcn_tables(asmc_ctx* ctx, const PcnDev& pd, hipStream_t st, double* rho_cell = nullptr, double rho = 0.0) {
    ctx->ptab_tag = 0;
    ASMC_LAUNCH(ctx, st, "k_pcn_pack", k_pcn_pack, dim3(1), dim3(256), 0, st, pd, ctx->d_ptab, rho_cell, rho);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static bool pcn_reg_supported(int d, size_t elem, const void* x) {
    return (d == 4 || d == 8 || d == 16 || d == 32) && (d * elem) % 16 == 0 && ((uintptr_t)x % 16) == 0 &&
           !getenv("ASMC_PCN_GENERIC");
}

template <typename T, int D, int NOISE, int MODE>
static int launch_pcn_reg(asmc_ctx* ctx, int64_t n, T* x, double* ll, double* lp, double* lq, const PcnDev& pd,
                          const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out, hipStream_t st) {
    constexpr int LDSROW = D * (int)sizeof(T) + 16;
    // waves per block: 160 KB of LDS hold floor(160K / tile) wave tiles; blocks of 4 waves waste the remainder
    // when the tile is large (d = 32 fp64: 17 KB tiles -> 2 blocks of 4 = 8 waves, but 9 single-wave blocks)
    constexpr size_t tile_bytes = (size_t)64 * LDSROW;
    static int wpb_env = getenv("ASMC_PCN_WPB") ? atoi(getenv("ASMC_PCN_WPB")) : 0;
    constexpr bool NO_LDS = MODE == PCN_Y_STEP_S || MODE == PCN_Y_STEP_TS || MODE == PCN_Y_STEP_SG ||
                            MODE == PCN_Y_STEP_TSG;  // coordinate-major state: registers only
    // kernels that draw the default noise keep its 6 KB of tables per BLOCK next to the tiles: four waves share them
    constexpr bool BM = NOISE == ASMC_NOISE_F64 && !(MODE == PCN_WHITEN || MODE == PCN_WHITEN_S || MODE == PCN_UNWHITEN ||
                                                     MODE == PCN_UNWHITEN_S || MODE == PCN_UNWHITEN_X || MODE == PCN_UNWHITEN_XS);
    const int wpb = wpb_env > 0 ? wpb_env : (BM || NO_LDS || (160 * 1024) / tile_bytes % 4 == 0 || tile_bytes * 12 <= 160 * 1024 ? 4 : 1);
    const size_t lds_bytes = NO_LDS ? 0 : (size_t)wpb * tile_bytes;
    const int64_t n_tiles = (n + 63) / 64;
    const int64_t grid64 = (n_tiles + wpb - 1) / wpb;
    if (grid64 > ASMC_PCN_MAX_GRID) {
        asmc_set_error("pcn: n=%lld exceeds the per-call block budget", (long long)n);
        return ASMC_ERR_UNSUPPORTED;
    }
    const int grid = (int)grid64;
    *grid_out = grid;
    auto kern = k_pcn_reg<T, D, NOISE, MODE>;
    static bool attr_set_dev[ASMC_MAX_DEVICES] = {false}; bool& attr_set = attr_set_dev[asmc_dev_slot(ctx)];  // per instantiation; the attribute call costs tens of microseconds
    if (lds_bytes > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    PcnScalars ps;
    ps.beta = pd.beta;
    ps.nu = pd.nu;
    ps.gam = pd.gam;
    ps.ys = pd.ys;
    ps.n_pad = pd.n_pad;
    ps.d_real = pd.d;
    ps.seed = pd.seed;
    ps.gid0 = pd.gid0;
    ps.bmtab = pd.bmtab;
    ps.tile_par = pd.tile_par;
    ps.d_noise = pd.d_noise;
    ps.c_ll = pd.ll.C;
    ps.c_lp = pd.lp.C;
    ps.c_lq = pd.lq.C;
    ASMC_LAUNCH(ctx, st, MODE == PCN_X_STEP ? "k_pcn_reg" : (MODE == PCN_Y_STEP || MODE == PCN_Y_STEP_S || MODE == PCN_Y_STEP_SG) ? "k_pcn_reg_y" : MODE == PCN_X_STEP_T ? "k_tpcn_reg" : (MODE == PCN_Y_STEP_T || MODE == PCN_Y_STEP_TS || MODE == PCN_Y_STEP_TSG) ? "k_tpcn_reg_y" : (MODE == PCN_WHITEN || MODE == PCN_WHITEN_S) ? "k_pcn_whiten" : (MODE >= PCN_X_PROPOSE && MODE <= PCN_X_PROPOSE_PAD_T) ? "k_pcn_propose_reg" : "k_pcn_unwhiten", kern, dim3(grid), dim3(wpb * 64), lds_bytes, st, n, x, ll, lp, lq, (const double*)ctx->d_ptab, ps,
                       rho_ptr, step, block_counts);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

template <typename T, int D, int NOISE, int MODE>
static int launch_pcn_reg_flow(asmc_ctx* ctx, int64_t n, T* y, T* x_prop, double* ll, double* lp, double* lq,
                               double* ll_new, double* lp_new, const double* lq_new, const PcnDev& pd,
                               const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                               hipStream_t st) {
    constexpr int LDSROW = D * (int)sizeof(T) + 16;
    constexpr size_t tile_bytes = (size_t)64 * LDSROW;
    constexpr bool NO_LDS = MODE == PCN_FLOW_ACCEPT_S || MODE == PCN_FLOW_ACCEPT_SJ;
    const int wpb = (NOISE == ASMC_NOISE_F64 || NO_LDS || (160 * 1024) / tile_bytes % 4 == 0 || tile_bytes * 12 <= 160 * 1024) ? 4 : 1;  // (f64 noise: 6 KB of tables per block)
    const size_t lds_bytes = NO_LDS ? 0 : (size_t)wpb * tile_bytes;
    const int64_t grid64 = ((n + 63) / 64 + wpb - 1) / wpb;
    if (grid64 > ASMC_PCN_MAX_GRID) {
        asmc_set_error("pcn: n=%lld exceeds the per-call block budget", (long long)n);
        return ASMC_ERR_UNSUPPORTED;
    }
    *grid_out = (int)grid64;
    auto kern = k_pcn_reg_flow<T, D, NOISE, MODE>;
    static bool attr_set_dev[ASMC_MAX_DEVICES] = {false}; bool& attr_set = attr_set_dev[asmc_dev_slot(ctx)];
    if (lds_bytes > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    PcnScalars ps;
    ps.beta = pd.beta;
    ps.nu = pd.nu;
    ps.gam = pd.gam;
    ps.ys = pd.ys;
    ps.n_pad = pd.n_pad;
    ps.d_real = pd.d;
    ps.seed = pd.seed;
    ps.gid0 = pd.gid0;
    ps.bmtab = pd.bmtab;
    ps.tile_par = pd.tile_par;
    ps.d_noise = pd.d_noise;
    ps.c_ll = pd.ll.C;
    ps.c_lp = pd.lp.C;
    ps.c_lq = pd.lq.C;
    ASMC_LAUNCH(ctx, st, (MODE == PCN_FLOW_PROPOSE || MODE == PCN_FLOW_PROPOSE_S || MODE == PCN_FLOW_PROPOSE_SX || MODE >= PCN_FLOW_PROPOSE_SXT_LOGIT) ? "k_pcn_flow_propose" : "k_pcn_flow_accept", kern, dim3((int)grid64),
                dim3(wpb * 64), lds_bytes, st, n, y, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, (const double*)ctx->d_ptab, ps,
                rho_ptr, step, block_counts);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

template <typename T, int MODE>
static int dispatch_pcn_reg_flow(asmc_ctx* ctx, int64_t n, T* y, T* x_prop, double* ll, double* lp, double* lq,
                                 double* ll_new, double* lp_new, const double* lq_new, const PcnDev& pd,
                                 const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                                 hipStream_t st) {
#define PCN_FLOW_CASE(DD)                                                                                              \
    if (pd.d == DD) {                                                                                                  \
        if (pd.noise == ASMC_NOISE_F32)                                                                                \
            return launch_pcn_reg_flow<T, DD, ASMC_NOISE_F32, MODE>(ctx, n, y, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, \
                                                                    pd, rho_ptr, step, block_counts, grid_out, st);   \
        return launch_pcn_reg_flow<T, DD, ASMC_NOISE_F64, MODE>(ctx, n, y, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, pd, \
                                                                rho_ptr, step, block_counts, grid_out, st);           \
    }
    PCN_FLOW_CASE(4)
    PCN_FLOW_CASE(8)
    PCN_FLOW_CASE(16)
    PCN_FLOW_CASE(32)
#undef PCN_FLOW_CASE
    asmc_set_error("pcn flow: d=%d has no register-resident kernel", pd.d);
    return ASMC_ERR_UNSUPPORTED;
}

template <typename T, int PHASE>
static int launch_pcn_step(asmc_ctx* ctx, int64_t n, T* x, double* ll, double* lp, double* lq, const PcnDev& pd,
                           const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                           T* x_prop, double* qf_old, double* qf_new, hipStream_t st) {
    const int rowbytes = pd.d * (int)sizeof(T);
    if (PHASE == 0 && (pd.dpad > pd.d || pcn_reg_supported(pd.d, sizeof(T), x))) {
        const int dk = pd.dpad > pd.d ? pd.dpad : pd.d;
        switch (dk * 64 + (pd.noise == ASMC_NOISE_F32 ? 32 : 0) + pd.mode) {  // register-resident specialisations
#define PCN_CASE2(DD, NZ, MD) \
    case DD * 64 + (NZ == ASMC_NOISE_F32 ? 32 : 0) + MD: \
        return launch_pcn_reg<T, DD, NZ, MD>(ctx, n, x, ll, lp, lq, pd, rho_ptr, step, block_counts, grid_out, st);
#define PCN_CASE(DD)                            \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_X_STEP)   \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_X_STEP)   \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_Y_STEP)   \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_Y_STEP)   \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_X_STEP_T) \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_X_STEP_T) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_Y_STEP_T) \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_Y_STEP_T) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_Y_STEP_S)  \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_Y_STEP_S)  \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_Y_STEP_TS) \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_Y_STEP_TS) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_Y_STEP_SG)  \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_Y_STEP_SG)  \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_Y_STEP_TSG) \
    PCN_CASE2(DD, ASMC_NOISE_F32, PCN_Y_STEP_TSG) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_WHITEN_S)  \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_UNWHITEN_S) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_UNWHITEN_XS) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_X_PROPOSE) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_X_PROPOSE_T) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_WHITEN)   \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_UNWHITEN) \
    PCN_CASE2(DD, ASMC_NOISE_F64, PCN_UNWHITEN_X)
#ifdef PCN_DIAG_ONLY_D32S  // diagnostic builds (generated-code experiments): the coordinate-major d = 32 steps alone
            PCN_CASE2(32, ASMC_NOISE_F64, PCN_Y_STEP_S)
            PCN_CASE2(32, ASMC_NOISE_F64, PCN_Y_STEP_TS)
#else
            PCN_CASE(4)
            PCN_CASE(8)
            PCN_CASE(16)
            PCN_CASE(32)
            PCN_CASE2(32, ASMC_NOISE_F64, PCN_X_PROPOSE_PAD)
            PCN_CASE2(32, ASMC_NOISE_F64, PCN_X_PROPOSE_PAD_T)
#endif
#undef PCN_CASE
#undef PCN_CASE2
            default: break;
        }
    }
    if (pd.mode >= PCN_WHITEN_S) {  // coordinate-major / propose modes exist as register kernels only
        asmc_set_error("pcn: no kernel for mode %d at d=%d, noise=%d", pd.mode, pd.d, pd.noise);
        return ASMC_ERR_UNSUPPORTED;
    }
    const size_t per_wave = (size_t)64 * lds_row_stride(rowbytes) + (size_t)64 * 8 * pd.d;
    size_t lds_bytes = 0;
    const int wpb = waves_for_lds(per_wave, &lds_bytes);
    if (per_wave > 160 * 1024 - 1024 - BM_TAB_N * 16) {
        asmc_set_error("pcn: d=%d needs %zu B of LDS per wave (unsupported)", pd.d, per_wave);
        return ASMC_ERR_UNSUPPORTED;
    }
    const int64_t n_tiles = (n + 63) / 64;
    int blocks_per_cu = (int)((160 * 1024) / (lds_bytes + BM_TAB_N * 16));
    blocks_per_cu = blocks_per_cu < 1 ? 1 : (blocks_per_cu > 8 ? 8 : blocks_per_cu);
    int cap = ctx->num_cu * blocks_per_cu * 2;  // two rounds of resident blocks, grid-stride over tiles
    if (cap > ASMC_MAX_BLOCKS) cap = ASMC_MAX_BLOCKS;
    const int grid = grid_for(n_tiles, wpb, cap);
    *grid_out = grid;
    const int vec = pick_vec(rowbytes, x, x_prop ? (const void*)x_prop : (const void*)x);
    auto launch = [&](auto kern) {
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        ASMC_LAUNCH(ctx, st, PHASE == 0 ? "k_pcn_step_generic" : "k_pcn_propose", kern, dim3(grid), dim3(wpb * 64), lds_bytes, st, n, x, ll, lp, lq, pd, rho_ptr, step,
                           block_counts, x_prop, qf_old, qf_new, wpb);
    };
    if (vec == 16)
        launch(k_pcn_step<T, 16, PHASE>);
    else if (vec == 8)
        launch(k_pcn_step<T, 8, PHASE>);
    else
        launch(k_pcn_step<T, 4, PHASE>);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static int check_mixture(const asmc_mixture& m) {
    ASMC_REQUIRE(m.n_components >= 1 && m.n_components <= ASMC_MAX_COMPONENTS, "mixture: bad component count");
    ASMC_REQUIRE(m.logw_dev && m.mu_dev && m.prec_dev, "mixture: null device pointer");
    return ASMC_OK;
}

// ---- any d <= 128 on the fast kernels: zero-padding to the next supported width ------------------------------------------
// The register-resident kernels exist for d in {4, 8, 16, 32}, the matrix-core kernels for d in {64, 128}; every other d used
// to fall to the generic LDS kernel (6-9x slower).  A d-dimensional problem is the same as the D-dimensional one with
// x = (x, 0), mu = (mu, 0), L = diag(L, I), zero precisions on the padded coordinates of every density component, and NO
// noise on the padded coordinates (PcnDev.d_noise: y' = 0 there, |y|^2 and the Student-t dimension stay d's) - the real
// coordinates see exactly the arithmetic of the unpadded specification (same noise words: they are keyed by coordinate).
// Cost: two copy passes over the rows per call (pad, un-pad) and N x D x s bytes of scratch, grown on demand.
static int pcn_pad_dim(int d) {
    const int widths[] = {4, 8, 16, 32, 64, 128};
    for (int D : widths)
        if (d <= D) return D;
    return 0;
}
template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_pad_rows(int64_t n, int d, int D, const T* __restrict__ x, T* __restrict__ xp) {
    const int64_t total = n * D, stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t e = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / D;
        const int j = (int)(e - i * D);
        xp[e] = j < d ? x[i * d + j] : (T)0;
    }
}
template <typename T>
__global__ __launch_bounds__(ASMC_BLOCK) void k_unpad_rows(int64_t n, int d, int D, const T* __restrict__ xp, T* __restrict__ x) {
    const int64_t total = n * d, stride = (int64_t)gridDim.x * ASMC_BLOCK;
    for (int64_t e = (int64_t)blockIdx.x * ASMC_BLOCK + threadIdx.x; e < total; e += stride) {
        const int64_t i = e / d;
        const int j = (int)(e - i * d);
        x[e] = xp[i * D + j];
    }
}
// padded tables: mu[D] | L[D, D] | Linv[D, D] | 3 x (mu[C_max, D] | prec[C_max, D]) for ll, lp, lq
__global__ __launch_bounds__(256) void k_pad_tables(int d, int D, PcnDev p, double* __restrict__ out) {
    double* mu = out;
    double* L = mu + D;
    double* Li = L + (size_t)D * D;
    for (int e = threadIdx.x; e < D; e += 256) mu[e] = e < d ? p.mu[e] : 0.0;
    for (int e = threadIdx.x; e < D * D; e += 256) {
        const int i = e / D, j = e - i * D;
        const bool in = i < d && j < d;
        L[e] = in ? p.L[i * d + j] : (i == j ? 1.0 : 0.0);
        Li[e] = in ? p.Linv[i * d + j] : (i == j ? 1.0 : 0.0);
    }
    const MixDev* mix[3] = {&p.ll, &p.lp, &p.lq};
    for (int k = 0; k < 3; k++) {
        double* m = Li + (size_t)D * D + (size_t)k * 2 * ASMC_MAX_COMPONENTS * D;
        double* pr = m + (size_t)ASMC_MAX_COMPONENTS * D;
        for (int e = threadIdx.x; e < mix[k]->C * D; e += 256) {
            const int c = e / D, j = e - c * D;
            m[e] = j < d ? mix[k]->mu[c * d + j] : 0.0;
            pr[e] = j < d ? mix[k]->prec[c * d + j] : 0.0;
        }
    }
}

extern "C" {

int asmc_gaussian_draw(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const double* mu, const double* sigma,
                       uint64_t seed, uint64_t gid0, uint32_t draw_id, void* x_out, double* lq_out,
                       asmc_stream stream) {
    ASMC_REQUIRE(ctx && mu && sigma && x_out, "null pointer");
    ASMC_REQUIRE(n > 0 && d > 0 && d <= ASMC_MAX_DIMS, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n * ((d + 3) / 4), ASMC_BLOCK * 2, ASMC_MAX_BLOCKS * 2);
    const int grid2 = grid_for(n, ASMC_BLOCK, ASMC_MAX_BLOCKS * 2);
    if (x_dtype == ASMC_F64) {
        ASMC_LAUNCH(ctx, st, "k_gaussian_draw<double>", k_gaussian_draw<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, mu, sigma,
                           (unsigned long long)seed, (unsigned long long)gid0, draw_id, (double*)x_out, (const double*)ctx->d_bmtab);
        ASMC_LAUNCH_CHECK();
        if (lq_out) ASMC_LAUNCH(ctx, st, "k_gaussian_logq<double>", k_gaussian_logq<double>, dim3(grid2), dim3(ASMC_BLOCK), 0, st, n, d, mu, sigma, (const double*)x_out, lq_out);
    } else {
        ASMC_LAUNCH(ctx, st, "k_gaussian_draw<float>", k_gaussian_draw<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, mu, sigma,
                           (unsigned long long)seed, (unsigned long long)gid0, draw_id, (float*)x_out, (const double*)ctx->d_bmtab);
        ASMC_LAUNCH_CHECK();
        if (lq_out) ASMC_LAUNCH(ctx, st, "k_gaussian_logq<float>", k_gaussian_logq<float>, dim3(grid2), dim3(ASMC_BLOCK), 0, st, n, d, mu, sigma, (const float*)x_out, lq_out);
    }
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

static int mixture_logpdf_impl(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const asmc_mixture* density,
                               double* out, const double* premap, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && density && out, "null pointer");
    ASMC_REQUIRE(n > 0 && d > 0 && d <= ASMC_MAX_DIMS, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    int rc = check_mixture(*density);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int elem = x_dtype == ASMC_F64 ? 8 : 4;
    const int rowbytes = d * elem;
    size_t lds_bytes = 0;
    const int wpb = waves_for_lds((size_t)64 * lds_row_stride(rowbytes), &lds_bytes);
    const int64_t n_tiles = (n + 63) / 64;
    int cap = ctx->num_cu * 4;
    if (cap > ASMC_MAX_BLOCKS) cap = ASMC_MAX_BLOCKS;
    const int grid = grid_for(n_tiles, wpb, cap);
    const int vec = pick_vec(rowbytes, x, x);
    const MixDev m = to_dev(*density);
    {
        const int pieces = rowbytes / 16;
        if (rowbytes % 16 == 0 && ((uintptr_t)x % 16) == 0 && pieces >= 1 && pieces <= 64 && (pieces & (pieces - 1)) == 0 &&
            m.C <= 4 && !getenv("ASMC_MIXTURE_TILED")) {
            int lg = 0;
            while ((1 << lg) < pieces) lg++;
            const int g = grid_for(n * pieces, ASMC_BLOCK, ctx->num_cu * 32);
            if (x_dtype == ASMC_F64) {
                if (m.C == 1)
                    ASMC_LAUNCH(ctx, st, "k_mixture_logpdf", (k_mixture_flat<double, 1>), dim3(g), dim3(ASMC_BLOCK), 0, st, n, d, lg,
                                (const uint4*)x, m, out, premap);
                else
                    ASMC_LAUNCH(ctx, st, "k_mixture_logpdf", (k_mixture_flat<double, 4>), dim3(g), dim3(ASMC_BLOCK), 0, st, n, d, lg,
                                (const uint4*)x, m, out, premap);
            } else {
                if (m.C == 1)
                    ASMC_LAUNCH(ctx, st, "k_mixture_logpdf", (k_mixture_flat<float, 1>), dim3(g), dim3(ASMC_BLOCK), 0, st, n, d, lg,
                                (const uint4*)x, m, out, premap);
                else
                    ASMC_LAUNCH(ctx, st, "k_mixture_logpdf", (k_mixture_flat<float, 4>), dim3(g), dim3(ASMC_BLOCK), 0, st, n, d, lg,
                                (const uint4*)x, m, out, premap);
            }
            ASMC_LAUNCH_CHECK();
            return ASMC_OK;
        }
    }
    if (premap) {
        asmc_set_error("asmc_mixture_logpdf_premap: rows must be a power-of-two number (<= 64) of 16-byte pieces, <= 4 components");
        return ASMC_ERR_UNSUPPORTED;
    }
    auto launch = [&](auto kern, auto xp) {
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        ASMC_LAUNCH(ctx, st, "k_mixture_logpdf", kern, dim3(grid), dim3(wpb * 64), lds_bytes, st, n, d, xp, m, out, wpb);
    };
    if (x_dtype == ASMC_F64) {
        const double* xp = (const double*)x;
        if (vec == 16) launch(k_mixture_logpdf<double, 16>, xp);
        else launch(k_mixture_logpdf<double, 8>, xp);
    } else {
        const float* xp = (const float*)x;
        if (vec == 16) launch(k_mixture_logpdf<float, 16>, xp);
        else if (vec == 8) launch(k_mixture_logpdf<float, 8>, xp);
        else launch(k_mixture_logpdf<float, 4>, xp);
    }
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_mixture_logpdf(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const asmc_mixture* density,
                        double* out, asmc_stream stream) {
    return mixture_logpdf_impl(ctx, n, d, x_dtype, x, density, out, nullptr, stream);
}

int asmc_mixture_logpdf_premap(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* premap_dev,
                               const asmc_mixture* density, double* out, asmc_stream stream) {
    ASMC_REQUIRE(premap_dev != nullptr, "null premap");
    return mixture_logpdf_impl(ctx, n, d, x_dtype, x, density, out, premap_dev, stream);
}

int asmc_colsum(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, double* sum_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && sum_host, "null pointer");
    ASMC_REQUIRE(n > 0 && d > 0 && d <= ctx->d_max && d <= ASMC_BLOCK, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    int grid = grid_for(n, (ASMC_BLOCK / d) * 16, ctx->gram_blocks);
    if (x_dtype == ASMC_F64)
        ASMC_LAUNCH(ctx, st, "k_colsum<double>", k_colsum<double>, dim3(grid), dim3(ASMC_BLOCK), ASMC_BLOCK * sizeof(double), st, n, d, (const double*)x, ctx->d_gram);
    else
        ASMC_LAUNCH(ctx, st, "k_colsum<float>", k_colsum<float>, dim3(grid), dim3(ASMC_BLOCK), ASMC_BLOCK * sizeof(double), st, n, d, (const float*)x, ctx->d_gram);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_reduce_columns", k_reduce_columns, dim3(d), dim3(64), 0, st, grid, d, (const double*)ctx->d_gram, ctx->d_small);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(double) * d, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    memcpy(sum_host, ctx->h_pinned, sizeof(double) * d);
    return ASMC_OK;
}

__global__ void k_center_from_sum(int d, const double* __restrict__ sum, double n, double* __restrict__ center) {
    const int j = threadIdx.x;
    if (j < d) center[j] = sum[j] / n;
}

// The reference Gaussian of a mutation from the population moments, on the device (smc/minipcn.py:75-84: mean and
// covariance of the particles; this repository's pCN specification whitens with the Cholesky factor): what the host did with
// numpy between two temperatures - cov = G / (n - 1), symmetrised; L = chol(cov + jitter * mean(diag) * I) with the jitter
// ladder 0, 1e-12, 1e-10, ... of twelve tries (denom = n - 1); Linv = L^-1 - in ONE block behind the Gram kernel, so the mutation's kernels
// follow without a host round trip (fetch, LAPACK, upload: ~0.2 ms of idle GPU per temperature).
// out = [mu (seg) | L (d x d, zeros above the diagonal) | pad to seg x d | Linv (d x d)], seg = 32 ceil(d / 32) doubles;
// status[0] = jitter tries used (0: none), -1: not factorable / not finite.
#define REF_THREADS 256
// Also the factorisation inside the device-side EM of the Student-t reference (asmc_student_fit): `sum` == NULL leaves the mean
// alone, `tab` (mu | Linv's lower triangle packed by rows) is what k_student_estep stages, `em` / `it`: the EM's state record -
// iterations behind the one that converged are skipped, a failed factorisation is recorded there.
__global__ __launch_bounds__(1024) void k_ref_factor(int d, const double* __restrict__ sum, const double* __restrict__ gram,
                                                           double n_mean, double denom, double* __restrict__ out,
                                                           double* __restrict__ status, double* __restrict__ tab,
                                                           double* __restrict__ em, int it) {
    extern __shared__ __align__(16) double s_a[];  // [d][d + 1]
    __shared__ double s_diag[128], s_rdiag[128];
    __shared__ double s_scale;
    if (em && (double)it > em[2]) return;  // (uniform: every thread reads the same cell)
    const int tid = threadIdx.x, ld = d + 1, seg = (d + 31) / 32 * 32;
    const int NT = (int)blockDim.x;  // (one wave for d <= 32 - free barriers - was measured: 63 us against 40 with four)
    double* o_mu = out;
    double* o_L = out + seg;
    double* o_Li = out + seg + (size_t)seg * d;
    if (sum)
        for (int j = tid; j < d; j += NT) {
            const double mj = sum[j] / n_mean;
            if (out) o_mu[j] = mj;
            if (tab) tab[j] = mj;
        }
    const double inv_denom = 1.0 / denom;
    int tries = -1;
    double jitter = 0.0;
    for (int attempt = 0; attempt < 12; attempt++) {
        __syncthreads();
        for (int e = tid; e < d * d; e += NT) {
            const int i = e / d, j = e - i * d;
            s_a[i * ld + j] = 0.5 * (gram[(size_t)i * d + j] * inv_denom + gram[(size_t)j * d + i] * inv_denom);
        }
        __syncthreads();
        if (attempt == 0) {
            if (tid == 0) {
                double t = 0.0;
                for (int j = 0; j < d; j++) t += s_a[j * ld + j];
                t /= (double)d;
                s_scale = (t > 0.0 && t < INFINITY) ? t : 1.0;
            }
        } else {
            for (int j = tid; j < d; j += NT) s_a[j * ld + j] += jitter * s_scale;
        }
        // right-looking Cholesky with ONE barrier per column: the trailing block takes A[i][k] -= A[i][j] A[k][j] / A[j][j]
        // (kept symmetric: both halves are updated); column j itself is left unscaled - it is not read again - and becomes
        // L[i][j] = A[i][j] / sqrt(A[j][j]) in the pass behind the loop
        bool ok = true;
        for (int j = 0; j < d; j++) {
            __syncthreads();
            const double p = s_a[j * ld + j];  // the same value in every thread: the test below is uniform
            if (!(p > 0.0 && p < INFINITY)) {
                ok = false;
                break;
            }
            double rp = __builtin_amdgcn_rcp(p);  // hardware reciprocal + two Newton steps: the division's chain is half of a column's latency
            rp = fma(fma(-p, rp, 1.0), rp, rp);
            rp = fma(fma(-p, rp, 1.0), rp, rp);
            // threads as a (NT / 32) x 32 patch walking the trailing block: no integer division per element
            for (int i = j + 1 + (tid >> 5); i < d; i += NT >> 5) {
                const double li = s_a[i * ld + j] * rp;
                for (int k = j + 1 + (tid & 31); k < d; k += 32) s_a[i * ld + k] = fma(-li, s_a[k * ld + j], s_a[i * ld + k]);
            }
        }
        if (ok) {
            tries = attempt;
            break;
        }
        jitter = jitter == 0.0 ? 1e-12 : jitter * 100.0;
    }
    __syncthreads();
    if (tid == 0) {
        if (status) status[0] = (double)tries;
        if (em && tries < 0) em[3] = -1.0;
    }
    if (tries < 0) {
        // no factor: poison (L, Linv) so that a mutation that runs before the host has looked at the status cannot use the
        // factors an earlier fit left in this slot - NaN proposals are rejected and counted, never silently accepted
        if (out)
            for (int e = tid; e < d * d; e += NT) o_L[e] = __builtin_nan(""), o_Li[e] = __builtin_nan("");
        if (tab)
            for (int e = tid; e < d * (d + 1) / 2; e += NT) tab[d + e] = __builtin_nan("");
        return;
    }
    for (int j = tid; j < d; j += NT) {
        const double sd = sqrt(s_a[j * ld + j]);
        s_diag[j] = sd, s_rdiag[j] = 1.0 / sd;
    }
    __syncthreads();
    for (int e = tid; e < d * d; e += NT) {
        const int i = e / d, j = e - i * d;
        double v = 0.0;
        if (j < i) v = s_a[i * ld + j] * s_rdiag[j];
        if (j == i) v = s_diag[j];
        if (out) o_L[e] = v;
        if (j < i) s_a[i * ld + j] = v;  // (the strict lower triangle: no other thread touches it in this pass)
    }
    __syncthreads();
    // Linv by forward substitution, every column at once and without a block barrier: a group of G lanes of one wave solves
    // L x = e_c for column c - the lanes split the dot product of a row (a single lane's chain of d^2 / 2 dependent LDS reads
    // was 30 of the kernel's 40 us at d = 32 and 520 us at d = 128) - and keeps x_i (i > c) in the FREE upper triangle, at
    // A[c][i], its own row; L is only read.  LDS operations of a wave complete in order: the lanes of a group see x_i
    // as soon as the instruction that wrote it has issued.
    {
        const int per = NT / d;
        const int G = per >= 8 ? 8 : per >= 4 ? 4 : per >= 2 ? 2 : 1;
        const int c = tid / G, q = tid % G;
        if (c < d) {
            const double xc = s_rdiag[c];
            for (int i = c + 1; i < d; i++) {
                double acc = q == 0 ? s_a[i * ld + c] * xc : 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
                int k = c + 1 + q;
                for (; k + 3 * G < i; k += 4 * G) {  // four independent chains: the LDS reads of a row pipeline
                    acc = fma(s_a[i * ld + k], s_a[c * ld + k], acc);
                    acc1 = fma(s_a[i * ld + k + G], s_a[c * ld + k + G], acc1);
                    acc2 = fma(s_a[i * ld + k + 2 * G], s_a[c * ld + k + 2 * G], acc2);
                    acc3 = fma(s_a[i * ld + k + 3 * G], s_a[c * ld + k + 3 * G], acc3);
                }
                for (; k < i; k += G) acc = fma(s_a[i * ld + k], s_a[c * ld + k], acc);
                acc = (acc + acc1) + (acc2 + acc3);
                for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                if (q == 0) s_a[c * ld + i] = -acc * s_rdiag[i];
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < d * d; e += NT) {
        const int i = e / d, j = e - i * d;
        const double v = j < i ? s_a[j * ld + i] : j == i ? s_rdiag[i] : 0.0;
        if (out) o_Li[e] = v;
        if (tab && j <= i) tab[d + i * (i + 1) / 2 + j] = v;
    }
}

// Column sums and the Gram matrix centred on sum / n_mean in ONE enqueue and one synchronisation (the reference fit of a
// temperature boundary: the centre never visits the host; same division, same kernels, same bits as asmc_colsum -> host
// division -> asmc_centered_gram).  _enqueue leaves both results on their way to pinned memory, _fetch waits for the stream
// and hands them out: a caller with other work on the stream (the importance step's chain) pays one synchronisation for all.
int asmc_mean_gram_enqueue(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, int64_t n_mean, int across_flags,
                           asmc_stream stream) {
    const int across_ranks = across_flags & ASMC_GRAM_ACROSS_RANKS;
    ASMC_REQUIRE(ctx && x, "null pointer");
    ASMC_REQUIRE(n > 0 && n_mean > 0 && d > 0 && d <= ctx->d_max && d <= 128, "bad sizes (gram supports d <= 128)");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(asmc_gram_mm_supported(d, x) && d <= ASMC_BLOCK && !getenv("ASMC_GRAM_GENERIC"),
                 "shape without the device-side path (asmc_mean_gram falls back to the two calls; across_ranks: merge on the host)");
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    const allreduce_fn allreduce = reinterpret_cast<allreduce_fn>(ctx->rccl_allreduce);
    const int nccl_f64 = 8, nccl_sum = 0;  // rccl.h: ncclFloat64, ncclSum
    ASMC_REQUIRE(!across_ranks || (allreduce && ctx->rccl_comm), "across_ranks needs asmc_set_rccl");
    hipStream_t st = as_stream(stream);
    int grid = grid_for(n, (ASMC_BLOCK / d) * 16, ctx->gram_blocks);
    // rows that asmc_gather has just written AND that the caller vouches for (ASMC_GRAM_FROM_GATHER: nothing has rewritten them
    // since - the library cannot see a caller's own kernels): their column-sum partials came with the gather, no pass over the rows
    const bool from_gather = (across_flags & ASMC_GRAM_FROM_GATHER) && ctx->cs_n == n && ctx->cs_x == x && ctx->cs_d == d &&
                             x_dtype == ASMC_F64;
    if (from_gather)
        grid = ctx->cs_grid;
    else if (x_dtype == ASMC_F64)
        ASMC_LAUNCH(ctx, st, "k_colsum<double>", k_colsum<double>, dim3(grid), dim3(ASMC_BLOCK), ASMC_BLOCK * sizeof(double), st, n, d, (const double*)x, ctx->d_gram);
    else
        ASMC_LAUNCH(ctx, st, "k_colsum<float>", k_colsum<float>, dim3(grid), dim3(ASMC_BLOCK), ASMC_BLOCK * sizeof(double), st, n, d, (const float*)x, ctx->d_gram);
    ASMC_LAUNCH_CHECK();
    // the results stay on the device in ctx->d_ref = {sums [128], Gram} (d_small / d_partials are every call's scratch):
    // asmc_reference_factor reads them there, asmc_mean_gram_fetch copies them out when a caller wants them on the host.  The
    // reductions write them there themselves, the all-reduces of a sharded run work on them in place, and the Gram kernel forms
    // the centre sum / n_mean itself: no launch sits between the passes and the collectives.
    ASMC_LAUNCH(ctx, st, "k_reduce_columns", k_reduce_columns, dim3(d), dim3(from_gather ? 256 : 64), 0, st, grid, d,
                (const double*)ctx->d_gram, ctx->d_small, ctx->d_ref, (double*)nullptr, (double)n_mean);
    ASMC_LAUNCH_CHECK();
    if (across_ranks && allreduce(ctx->d_ref, ctx->d_ref, (size_t)d, nccl_f64, nccl_sum, ctx->rccl_comm, st) != 0) {
        asmc_set_error("asmc_mean_gram: ncclAllReduce failed");
        return ASMC_ERR_ARG;
    }
    int ggrid = 0;
    int rc = asmc_gram_mm_launch(ctx, n, d, x_dtype, x, ctx->d_ref, &ggrid, st, ctx->d_ref + 128, (double)n_mean);
    if (rc) return rc;
    if (across_ranks && allreduce(ctx->d_ref + 128, ctx->d_ref + 128, (size_t)d * d, nccl_f64, nccl_sum, ctx->rccl_comm, st) != 0) {
        asmc_set_error("asmc_mean_gram: ncclAllReduce failed");
        return ASMC_ERR_ARG;
    }
    ctx->gram_pending_d = d;
    return ASMC_OK;
}

int asmc_mean_gram_fetch(asmc_ctx* ctx, int d, double* sum_host, double* gram_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && sum_host && gram_host, "null pointer");
    ASMC_REQUIRE(ctx->gram_pending_d == d && d > 0, "no asmc_mean_gram_enqueue of this d is pending");
    ASMC_HIP(hipMemcpyAsync(ctx->h_gram, ctx->d_ref, sizeof(double) * (128 + (size_t)d * d), hipMemcpyDeviceToHost, as_stream(stream)));
    ASMC_HIP(hipStreamSynchronize(as_stream(stream)));
    memcpy(sum_host, ctx->h_gram, sizeof(double) * d);
    memcpy(gram_host, ctx->h_gram + 128, sizeof(double) * d * d);
    ctx->gram_pending_d = 0;
    return ASMC_OK;
}

}  // extern "C"
// Every factorisation request reads its status back into a pinned cell OF ITS OWN (a ring indexed by a generation counter): the
// host writes the "not yet known" sentinel into a cell that no copy still in flight targets - an earlier request's late copy
// lands in its own cell and cannot be mistaken for this request's status.  (The ring is deeper than the requests a caller can
// have in flight between two synchronisations: one per temperature.)
#define REF_STATUS_CELL0 8010
#define REF_STATUS_CELLS 16
static int ref_status_request(asmc_ctx* ctx, const double* d_status, hipStream_t st) {
    ctx->ref_status_gen++;
    double* cell = ctx->h_pinned + REF_STATUS_CELL0 + ctx->ref_status_gen % REF_STATUS_CELLS;
    *cell = -2.0;
    ASMC_HIP(hipMemcpyAsync(cell, d_status, sizeof(double), hipMemcpyDeviceToHost, st));
    return ASMC_OK;
}

int asmc_ref_factor_launch(asmc_ctx* ctx, int d, const double* sum, const double* gram, double n_mean, double denom, double* out,
                           double* status, double* tab, double* em, int it, hipStream_t st) {
    const size_t lds = sizeof(double) * (size_t)d * (d + 1);
    static size_t attr_lds_dev[ASMC_MAX_DEVICES] = {0}; size_t& attr_lds = attr_lds_dev[asmc_dev_slot(ctx)];
    if (lds > 64 * 1024 && lds > attr_lds) {
        ASMC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ref_factor), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    // d <= 32 is bound by the per-column latency whatever the block (38-41 us from 256 to 1024 threads); d = 128 by the trailing
    // updates: 683 us with 256 threads, 516 with 1024
    static const int ref_env = getenv("ASMC_REF_THREADS") ? atoi(getenv("ASMC_REF_THREADS")) : 0;
    const int ref_threads = ref_env > 0 ? ref_env : (d <= 32 ? REF_THREADS : 1024);
    ASMC_LAUNCH(ctx, st, "k_ref_factor", k_ref_factor, dim3(1), dim3(ref_threads), lds, st, d, sum, gram, n_mean, denom, out,
                status, tab, em, it);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}
extern "C" {

// (mu, L, Linv) of the reference Gaussian from the moments of the pending asmc_mean_gram_enqueue (consumed: no fetch
// follows) or, with sum_host / gram_host, from moments the caller merged on the host (uploaded first): k_ref_factor on the
// stream.  The status lands in pinned memory behind it; asmc_reference_factor_status reads it after the caller's next
// synchronisation of the stream.
int asmc_reference_factor(asmc_ctx* ctx, int d, int64_t n_mean, int64_t n_cov, const double* sum_host, const double* gram_host,
                          double* out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && out_dev, "null pointer");
    ASMC_REQUIRE(d > 0 && d <= 128 && n_mean > 0 && n_cov > 0, "bad sizes (d <= 128)");
    ASMC_REQUIRE((sum_host == nullptr) == (gram_host == nullptr), "sum_host and gram_host come together");
    hipStream_t st = as_stream(stream);
    if (gram_host) {
        ASMC_REQUIRE(ctx->gram_pending_d == 0, "an asmc_mean_gram_enqueue is pending: its results would be overwritten");
        ASMC_HIP(hipStreamSynchronize(st));  // (the pinned staging may still feed an earlier copy)
        memcpy(ctx->h_gram, sum_host, sizeof(double) * d);
        memcpy(ctx->h_gram + 128, gram_host, sizeof(double) * d * d);
        ASMC_HIP(hipMemcpyAsync(ctx->d_ref, ctx->h_gram, sizeof(double) * (128 + (size_t)d * d), hipMemcpyHostToDevice, st));
    } else {
        ASMC_REQUIRE(ctx->gram_pending_d == d, "no asmc_mean_gram_enqueue of this d is pending");
        ctx->gram_pending_d = 0;
    }
    double* d_status = ctx->d_small + 2300;
    const int rc = asmc_ref_factor_launch(ctx, d, ctx->d_ref, ctx->d_ref + 128, (double)n_mean, (double)(n_cov - 1 > 1 ? n_cov - 1 : 1),
                                          out_dev, d_status, nullptr, nullptr, 0, st);
    if (rc) return rc;
    return ref_status_request(ctx, d_status, st);
}

// The sharded form of the same fit without a host round trip: column sums and the centred Gram matrix into the CALLER's device
// buffers (the caller sums each over the ranks with its own stream-ordered all-reduce), then the factorisation from them.
__global__ __launch_bounds__(256) void k_copy_doubles(int n, const double* __restrict__ src, double* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) dst[e] = src[e];
}

int asmc_colsum_dev(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, int from_gather_flag, double* sum_dev,
                    asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && sum_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && d > 0 && d <= ctx->d_max && d <= ASMC_BLOCK, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    int grid = grid_for(n, (ASMC_BLOCK / d) * 16, ctx->gram_blocks);
    const bool from_gather = from_gather_flag && ctx->cs_n == n && ctx->cs_x == x && ctx->cs_d == d && x_dtype == ASMC_F64;  // (see asmc_mean_gram_enqueue)
    if (from_gather)
        grid = ctx->cs_grid;
    else if (x_dtype == ASMC_F64)
        ASMC_LAUNCH(ctx, st, "k_colsum<double>", k_colsum<double>, dim3(grid), dim3(ASMC_BLOCK), ASMC_BLOCK * sizeof(double), st, n, d, (const double*)x, ctx->d_gram);
    else
        ASMC_LAUNCH(ctx, st, "k_colsum<float>", k_colsum<float>, dim3(grid), dim3(ASMC_BLOCK), ASMC_BLOCK * sizeof(double), st, n, d, (const float*)x, ctx->d_gram);
    ASMC_LAUNCH_CHECK();
    ASMC_LAUNCH(ctx, st, "k_reduce_columns", k_reduce_columns, dim3(d), dim3(from_gather ? 256 : 64), 0, st, grid, d, (const double*)ctx->d_gram, sum_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_centered_gram_dev(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* sum_dev, int64_t n_mean,
                           double* gram_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && sum_dev && gram_dev, "null pointer");
    ASMC_REQUIRE(n > 0 && n_mean > 0 && d > 0 && d <= ctx->d_max && d <= 128, "bad sizes (gram supports d <= 128)");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    if (!asmc_gram_mm_supported(d, x) || getenv("ASMC_GRAM_GENERIC")) {
        asmc_set_error("asmc_centered_gram_dev: shape without the matrix-core Gram kernel (d in {32, 64, 128}, 16-byte aligned rows)");
        return ASMC_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    double* d_center = ctx->d_small + 2048;
    ASMC_LAUNCH(ctx, st, "k_center_from_sum", k_center_from_sum, dim3(1), dim3(128), 0, st, d, sum_dev, (double)n_mean, d_center);
    ASMC_LAUNCH_CHECK();
    int ggrid = 0;
    const int rc = asmc_gram_mm_launch(ctx, n, d, x_dtype, x, d_center, &ggrid, st, nullptr, 0.0);
    if (rc) return rc;
    ASMC_LAUNCH(ctx, st, "k_copy_doubles", k_copy_doubles, dim3((d * d + 255) / 256), dim3(256), 0, st, d * d, (const double*)ctx->d_partials,
                gram_dev);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_reference_factor_dev(asmc_ctx* ctx, int d, int64_t n_mean, int64_t n_cov, const double* sum_dev, const double* gram_dev,
                              double* out_dev, asmc_stream stream) {
    ASMC_REQUIRE(ctx && sum_dev && gram_dev && out_dev, "null pointer");
    ASMC_REQUIRE(d > 0 && d <= 128 && n_mean > 0 && n_cov > 0, "bad sizes (d <= 128)");
    hipStream_t st = as_stream(stream);
    double* d_status = ctx->d_small + 2300;
    const int rc = asmc_ref_factor_launch(ctx, d, sum_dev, gram_dev, (double)n_mean, (double)(n_cov - 1 > 1 ? n_cov - 1 : 1), out_dev,
                                          d_status, nullptr, nullptr, 0, st);
    if (rc) return rc;
    return ref_status_request(ctx, d_status, st);
}

int asmc_reference_factor_status(asmc_ctx* ctx, int* status_host) {
    ASMC_REQUIRE(ctx && status_host, "null pointer");
    *status_host = (int)ctx->h_pinned[REF_STATUS_CELL0 + ctx->ref_status_gen % REF_STATUS_CELLS];  // -2: the stream has not been synchronised since asmc_reference_factor
    return ASMC_OK;
}

// the request a caller has just made, and the status of one particular request: a mutation asks about the factorisation
// that served IT - the next temperature's may already be on the stream behind it (asmc_reference_factor_status reads the latest
// request's cell: -2 until that one has run)
int64_t asmc_reference_factor_generation(asmc_ctx* ctx) { return ctx ? (int64_t)ctx->ref_status_gen : -1; }
int asmc_reference_factor_status_of(asmc_ctx* ctx, int64_t generation, int* status_host) {
    ASMC_REQUIRE(ctx && status_host, "null pointer");
    ASMC_REQUIRE(generation > 0 && (uint64_t)generation <= ctx->ref_status_gen &&
                     ctx->ref_status_gen - (uint64_t)generation < REF_STATUS_CELLS,
                 "no such factorisation request (or more than 15 requests ago)");
    *status_host = (int)ctx->h_pinned[REF_STATUS_CELL0 + (unsigned)generation % REF_STATUS_CELLS];
    return ASMC_OK;
}

int asmc_mean_gram(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, int64_t n_mean, int across_ranks,
                   double* sum_host, double* gram_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && sum_host && gram_host, "null pointer");
    ASMC_REQUIRE(n > 0 && n_mean > 0 && d > 0 && d <= ctx->d_max && d <= 128, "bad sizes (gram supports d <= 128)");
    if (!across_ranks && (!(asmc_gram_mm_supported(d, x) && d <= ASMC_BLOCK) || getenv("ASMC_GRAM_GENERIC"))) {
        int rc = asmc_colsum(ctx, n, d, x_dtype, x, sum_host, stream);  // shapes without the fp64-MFMA Gram kernel
        if (rc) return rc;
        double center[128];
        for (int j = 0; j < d; j++) center[j] = sum_host[j] / (double)n_mean;
        return asmc_centered_gram(ctx, n, d, x_dtype, x, center, gram_host, stream);
    }
    const int rc = asmc_mean_gram_enqueue(ctx, n, d, x_dtype, x, n_mean, across_ranks, stream);
    if (rc) return rc;
    return asmc_mean_gram_fetch(ctx, d, sum_host, gram_host, stream);
}

int asmc_centered_gram(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* center_host,
                       double* gram_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && center_host && gram_host, "null pointer");
    ASMC_REQUIRE(n > 0 && d > 0 && d <= ctx->d_max && d <= 128, "bad sizes (gram supports d <= 128)");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    ASMC_HIP(hipStreamSynchronize(st));
    memcpy(ctx->h_pinned + 2048, center_host, sizeof(double) * d);
    double* d_center = ctx->d_small + 2048;
    ASMC_HIP(hipMemcpyAsync(d_center, ctx->h_pinned + 2048, sizeof(double) * d, hipMemcpyHostToDevice, st));
    const size_t elem = x_dtype == ASMC_F64 ? 8 : 4;
    const int rowbytes = (int)(d * elem);
    double* d_out = ctx->d_partials;
    if (asmc_gram_mm_supported(d, x) && ctx->d_max >= d && !getenv("ASMC_GRAM_GENERIC")) {  // fp64 MFMA (asmc_pcn_mm.hip)
        int grid = 0;
        int rc = asmc_gram_mm_launch(ctx, n, d, x_dtype, x, d_center, &grid, st, nullptr, 0.0);
        if (rc) return rc;
        ASMC_HIP(hipMemcpyAsync(gram_host, d_out, sizeof(double) * d * d, hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipStreamSynchronize(st));
        return ASMC_OK;
    }
    {
        // any other d <= 128: a zero-padded copy of the rows (centre padded with zeros) through the matrix-core kernel of the next
        // width >= 32; the d x d corner of its result is the answer (k_gram_rb took 1 ms at d = 48 and 8.5 ms at d = 100 per call)
        const int D = pcn_pad_dim(d) < 32 ? 32 : pcn_pad_dim(d);
        if (D > 0 && D != d && D <= (ctx->d_max_pad < 32 ? 32 : ctx->d_max_pad) && !getenv("ASMC_GRAM_GENERIC") && !getenv("ASMC_PCN_NOPAD")) {
            const size_t need = (size_t)n * D * elem;
            if (need > ctx->xpad_bytes) {
                ASMC_HIP(hipStreamSynchronize(st));
                if (ctx->d_xpad) (void)hipFree(ctx->d_xpad);
                ctx->d_xpad = nullptr;
                ctx->xpad_bytes = 0;
                if (hipMalloc(&ctx->d_xpad, need) != hipSuccess) {
                    (void)hipGetLastError();
                    asmc_set_error("centered_gram: no device memory for the zero-padded copy of the rows (%zu bytes)", need);
                    return ASMC_ERR_NOMEM;
                }
                ctx->xpad_bytes = need;
            }
            for (int j = d; j < D; j++) ctx->h_pinned[2048 + j] = 0.0;
            ASMC_HIP(hipMemcpyAsync(d_center, ctx->h_pinned + 2048, sizeof(double) * D, hipMemcpyHostToDevice, st));
            const int pg = grid_for(n * D, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
            if (elem == 8)
                ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<double>, dim3(pg), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)x, (double*)ctx->d_xpad);
            else
                ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<float>, dim3(pg), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)x, (float*)ctx->d_xpad);
            ASMC_LAUNCH_CHECK();
            int grid = 0;
            int rc = asmc_gram_mm_launch(ctx, n, D, x_dtype, ctx->d_xpad, d_center, &grid, st, nullptr, 0.0);
            if (rc) return rc;
            ASMC_HIP(hipMemcpy2DAsync(gram_host, sizeof(double) * d, d_out, sizeof(double) * D, sizeof(double) * d, d,
                                      hipMemcpyDeviceToHost, st));
            ASMC_HIP(hipStreamSynchronize(st));
            return ASMC_OK;
        }
    }
    if (rowbytes % 16 == 0 && ((uintptr_t)x % 16) == 0 && d <= 128) {
        // register-blocked kernel: BLK = 4 (quadrant 32) for d <= 32, else BLK = 8 (quadrant 64)
        const int blk = d <= 32 ? 4 : 8;
        const int Q = 8 * blk;
        const int nq = (d + Q - 1) / Q;
        const int dpad = nq * Q;
        const int wpb = d > 64 ? 2 : ASMC_BLOCK / 64;  // d = 128: two waves per block keep the tiles inside 160 KB of LDS
        const size_t lds = (size_t)wpb * 64 * lds_row_stride(rowbytes);
        const size_t lds_red = (size_t)wpb * Q * Q * sizeof(double);
        const size_t lds_bytes = lds > lds_red ? lds : lds_red;
        int cap = (int)(((size_t)ctx->gram_blocks * ctx->d_max * ctx->d_max) / ((size_t)dpad * dpad));
        if (cap > ctx->num_cu * 2) cap = ctx->num_cu * 2;
        if (cap < 1) {
            asmc_set_error("centered_gram: ctx d_max=%d too small for d=%d", ctx->d_max, d);
            return ASMC_ERR_ARG;
        }
        const int grid = grid_for((n + 63) / 64, wpb, cap);
        auto launch = [&](auto kern, auto xp) {
            if (lds_bytes > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            ASMC_LAUNCH(ctx, st, "k_gram_rb", kern, dim3(grid, nq * nq), dim3(wpb * 64), lds_bytes, st, n, d, xp,
                        (const double*)d_center, ctx->d_gram, nq);
        };
        if (x_dtype == ASMC_F64) {
            if (blk == 4) launch(k_gram_rb<double, 4>, (const double*)x);
            else launch(k_gram_rb<double, 8>, (const double*)x);
        } else {
            if (blk == 4) launch(k_gram_rb<float, 4>, (const float*)x);
            else launch(k_gram_rb<float, 8>, (const float*)x);
        }
        ASMC_LAUNCH_CHECK();
        ASMC_LAUNCH(ctx, st, "k_reduce_columns", k_reduce_columns, dim3(dpad * dpad < 1024 ? dpad * dpad : 1024), dim3(64), 0, st, grid,
                    dpad * dpad, (const double*)ctx->d_gram, d_out);
        ASMC_LAUNCH_CHECK();
        // strip the padding while copying back
        ASMC_HIP(hipMemcpy2DAsync(gram_host, sizeof(double) * d, d_out, sizeof(double) * dpad, sizeof(double) * d, d,
                                  hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipStreamSynchronize(st));
        return ASMC_OK;
    }
    ASMC_REQUIRE(d <= 64, "centered_gram: unaligned rows are supported for d <= 64 only");
    const int grid = grid_for(n, 64 * 8, ctx->gram_blocks);
    const size_t lds = sizeof(double) * 64 * (d + 1);
    if (x_dtype == ASMC_F64)
        ASMC_LAUNCH(ctx, st, "k_gram<double>", k_gram<double>, dim3(grid), dim3(ASMC_BLOCK), lds, st, n, d, (const double*)x, (const double*)d_center, ctx->d_gram);
    else
        ASMC_LAUNCH(ctx, st, "k_gram<float>", k_gram<float>, dim3(grid), dim3(ASMC_BLOCK), lds, st, n, d, (const float*)x, (const double*)d_center, ctx->d_gram);
    ASMC_LAUNCH_CHECK();
    // d*d <= 4096 doubles: reduce into d_partials, then read back
    ASMC_LAUNCH(ctx, st, "k_reduce_columns", k_reduce_columns, dim3(d * d < 1024 ? d * d : 1024), dim3(64), 0, st, grid, d * d, (const double*)ctx->d_gram, d_out);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemcpyAsync(gram_host, d_out, sizeof(double) * d * d, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    return ASMC_OK;
}

// coordinate-major scratch for the whitened state of one mutation (grown on demand, kept for the life of the ctx);
// false when the device has no room for it (callers then stay on the in-place row-major path)
static bool pcn_ensure_ysoa(asmc_ctx* ctx, int64_t n, int d, int x_dtype, PcnDev& pd, hipStream_t st, bool with_scratch = false) {
    if (getenv("ASMC_PCN_AOS")) return false;
    const int64_t n_pad = ((n + 63) / 64) * 64;
    const size_t one = (size_t)n_pad * d * (x_dtype == ASMC_F64 ? 8 : 4);
    if (one >= (1ULL << 32)) return false;  // the kernels address the buffer through one 32-bit-offset descriptor
    const size_t need = with_scratch ? 2 * one : one;  // fused flow step: y' scratch of the same layout behind the state
    if (need > ctx->ysoa_bytes) {
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        if (ctx->d_ysoa) (void)hipFree(ctx->d_ysoa);
        ctx->d_ysoa = nullptr;
        ctx->ysoa_bytes = 0;
        if (hipMalloc(&ctx->d_ysoa, need) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        ctx->ysoa_bytes = need;
        // the fused flow step reads whole 64-particle tiles: the slots behind a ragged last tile must hold finite numbers
        if (hipMemsetAsync(ctx->d_ysoa, 0, need, st) != hipSuccess) return false;
    }
    pd.ys = ctx->d_ysoa;
    pd.n_pad = n_pad;
    return true;
}

int asmc_pcn_set_count_hook(asmc_ctx* ctx, asmc_count_hook hook, void* user, int64_t* cell_dev, int64_t n_global) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(hook == nullptr || (cell_dev != nullptr && n_global > 0), "hook needs a device cell and n_global > 0");
    ctx->count_hook = hook;
    ctx->count_hook_user = user;
    ctx->count_cell = reinterpret_cast<long long*>(cell_dev);
    ctx->count_n_global = n_global;
    ctx->count_cells = 1;  // (asmc_pcn_set_count_cells widens it)
    return ASMC_OK;
}

int asmc_pcn_set_count_cells(asmc_ctx* ctx, int n_cells) {
    ASMC_REQUIRE(ctx != nullptr && n_cells >= 1 && n_cells <= ASMC_MAX_COUNT_CELLS, "n_cells out of range");
    ctx->count_cells = n_cells;
    return ASMC_OK;
}

// The reference checks the carried log q for NaN after every mutation (smc/minipcn.py: "Log proposal contains NaN values").
// The count rides on the mutation call's own read-back instead of a call and a synchronisation of its own: NaN / inf counts
// of lq -> h_pinned[8002 .. 8003], read by asmc_pcn_lq_nan after the call's synchronisation.
static int pcn_enqueue_lq_check(asmc_ctx* ctx, int64_t n, const double* lq, hipStream_t st) {
    const int rc = asmc_count_nonfinite_enqueue(ctx, n, lq, st);
    if (rc) return rc;
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned + 8002, asmc_count_slot(ctx), sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, st));
    return ASMC_OK;
}

// the exchange issued by the library: RCCL's all-reduce on the step kernels' own stream (include/asmc.h)
typedef int (*rccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static int rccl_count_hook(void* user, asmc_stream stream) {
    asmc_ctx* ctx = static_cast<asmc_ctx*>(user);
    const int nccl_int64 = 4, nccl_sum = 0;  // rccl.h: ncclInt64, ncclSum
    return reinterpret_cast<rccl_allreduce_fn>(ctx->rccl_allreduce)(ctx->count_cell, ctx->count_cell, (size_t)ctx->count_cells, nccl_int64, nccl_sum,
                                                                    ctx->rccl_comm, reinterpret_cast<hipStream_t>(stream));
}

int asmc_set_rccl(asmc_ctx* ctx, void* allreduce_fn, void* nccl_comm) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE((allreduce_fn == nullptr) == (nccl_comm == nullptr), "function and communicator come together");
    ctx->rccl_allreduce = allreduce_fn;
    ctx->rccl_comm = nccl_comm;
    return ASMC_OK;
}

int asmc_set_rccl_allgather(asmc_ctx* ctx, void* allgather_fn) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ctx->rccl_allgather = allgather_fn;
    return ASMC_OK;
}

// equal-sized all-gather of 8-byte elements on the library's communicator and stream (the small exchanges of owner-layout
// resampling: evidence partials, chain states, kept counts)
int asmc_rccl_all_gather(asmc_ctx* ctx, const void* send_dev, void* recv_dev, int64_t count, int is_int64, asmc_stream stream) {
    ASMC_REQUIRE(ctx && send_dev && recv_dev && count > 0, "bad arguments");
    ASMC_REQUIRE(ctx->rccl_allgather && ctx->rccl_comm, "asmc_set_rccl / asmc_set_rccl_allgather first");
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
    const int dt = is_int64 ? 4 : 8;  // rccl.h: ncclInt64, ncclFloat64
    if (reinterpret_cast<allgather_fn>(ctx->rccl_allgather)(send_dev, recv_dev, (size_t)count, dt, ctx->rccl_comm,
                                                            reinterpret_cast<hipStream_t>(stream)) != 0) {
        asmc_set_error("asmc_rccl_all_gather: ncclAllGather failed");
        return ASMC_ERR_ARG;
    }
    return ASMC_OK;
}

int asmc_pcn_set_count_rccl(asmc_ctx* ctx, int64_t* cell_dev, int64_t n_global) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    if (cell_dev == nullptr) return asmc_pcn_set_count_hook(ctx, nullptr, nullptr, nullptr, 0);
    ASMC_REQUIRE(ctx->rccl_allreduce && ctx->rccl_comm, "asmc_set_rccl first");
    return asmc_pcn_set_count_hook(ctx, rccl_count_hook, ctx, cell_dev, n_global);
}

}  // extern "C" (templates below)

static int pcn_mutate_impl(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq, const asmc_pcn_params* prm,
                           int n_steps, uint32_t step0, double* rho_inout_host, int64_t* n_accept_host, double* rho_hist_host,
                           asmc_stream stream, int d_noise);

static int pcn_mutate_padded(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq, const asmc_pcn_params* prm,
                             int D, int n_steps, uint32_t step0, double* rho_inout_host, int64_t* n_accept_host,
                             double* rho_hist_host, asmc_stream stream) {
    hipStream_t st = as_stream(stream);
    const int d = prm->d;
    const size_t es = prm->x_dtype == ASMC_F64 ? 8 : 4;
    const size_t tab_doubles = (size_t)D + 2 * (size_t)D * D + 3 * 2 * (size_t)ASMC_MAX_COMPONENTS * D;
    const size_t tab_bytes = ((tab_doubles * 8 + 255) / 256) * 256;
    const size_t need = tab_bytes + (size_t)n * D * es;
    if (need > ctx->xpad_bytes) {
        ASMC_HIP(hipStreamSynchronize(st));
        if (ctx->d_xpad) (void)hipFree(ctx->d_xpad);
        ctx->d_xpad = nullptr;
        ctx->xpad_bytes = 0;
        if (hipMalloc(&ctx->d_xpad, need) != hipSuccess) {
            (void)hipGetLastError();
            asmc_set_error("pcn: no device memory for the zero-padded copy of the state (%zu bytes)", need);
            return ASMC_ERR_NOMEM;
        }
        ctx->xpad_bytes = need;
    }
    double* tab = reinterpret_cast<double*>(ctx->d_xpad);
    void* xp = reinterpret_cast<char*>(ctx->d_xpad) + tab_bytes;
    PcnDev src;
    memset(&src, 0, sizeof(src));
    src.mu = prm->mu_dev, src.L = prm->L_dev, src.Linv = prm->Linv_dev;
    src.ll = to_dev(prm->log_likelihood), src.lp = to_dev(prm->log_prior), src.lq = to_dev(prm->log_q);
    ASMC_LAUNCH(ctx, st, "k_pad_tables", k_pad_tables, dim3(1), dim3(256), 0, st, d, D, src, tab);
    ASMC_LAUNCH_CHECK();
    const int grid = grid_for(n * D, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
    if (es == 8)
        ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)x, (double*)xp);
    else
        ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)x, (float*)xp);
    ASMC_LAUNCH_CHECK();
    asmc_pcn_params p2 = *prm;
    p2.d = D;
    p2.mu_dev = tab;
    p2.L_dev = tab + D;
    p2.Linv_dev = p2.L_dev + (size_t)D * D;
    asmc_mixture* mix[3] = {&p2.log_likelihood, &p2.log_prior, &p2.log_q};
    for (int k = 0; k < 3; k++) {
        mix[k]->mu_dev = p2.Linv_dev + (size_t)D * D + (size_t)k * 2 * ASMC_MAX_COMPONENTS * D;
        mix[k]->prec_dev = mix[k]->mu_dev + (size_t)ASMC_MAX_COMPONENTS * D;
    }
    int rc = pcn_mutate_impl(ctx, n, xp, ll, lp, lq, &p2, n_steps, step0, rho_inout_host, n_accept_host, rho_hist_host, stream, d);
    if (rc) return rc;
    if (es == 8)
        ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)xp, (double*)x);
    else
        ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)xp, (float*)x);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

extern "C" {

int asmc_pcn_mutate(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq,
                    const asmc_pcn_params* prm, int n_steps, uint32_t step0, double* rho_inout_host,
                    int64_t* n_accept_host, double* rho_hist_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && prm, "null pointer");
    ASMC_REQUIRE(prm->d > 0 && prm->d <= ASMC_MAX_DIMS, "bad d");
    const int D = pcn_pad_dim(prm->d);
    const bool fast_as_is = prm->d == D;
    if (!fast_as_is && D > 0 && D <= ctx->d_max_pad && !getenv("ASMC_PCN_GENERIC") && !getenv("ASMC_PCN_NOPAD")) {
        ASMC_REQUIRE(ll && lp && lq && rho_inout_host && n_accept_host, "null pointer");
        ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
        ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
        ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference-Gaussian pointer");
        int rc = check_mixture(prm->log_likelihood);
        if (!rc) rc = check_mixture(prm->log_prior);
        if (!rc) rc = check_mixture(prm->log_q);
        if (rc) return rc;
        return pcn_mutate_padded(ctx, n, x, ll, lp, lq, prm, D, n_steps, step0, rho_inout_host, n_accept_host, rho_hist_host, stream);
    }
    return pcn_mutate_impl(ctx, n, x, ll, lp, lq, prm, n_steps, step0, rho_inout_host, n_accept_host, rho_hist_host, stream, 0);
}

static int pcn_mutate_impl(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq, const asmc_pcn_params* prm,
                           int n_steps, uint32_t step0, double* rho_inout_host, int64_t* n_accept_host, double* rho_hist_host,
                           asmc_stream stream, int d_noise) {
    ASMC_REQUIRE(ctx && x && ll && lp && lq && prm && rho_inout_host && n_accept_host, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    ASMC_REQUIRE(n_steps >= 1 && n_steps <= ASMC_MAX_PCN_STEPS, "n_steps out of range (<= 2048 per call)");
    ASMC_REQUIRE(prm->d > 0 && prm->d <= ASMC_MAX_DIMS, "bad d");
    ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference-Gaussian pointer");
    ASMC_REQUIRE(*rho_inout_host > 0.0 && *rho_inout_host <= 1.0, "rho must be in (0, 1]");
    ASMC_REQUIRE(!(prm->nu > 0.0) || prm->nu >= 1.0, "nu must be >= 1 (or <= 0 for the Gaussian reference)");
    int rc = check_mixture(prm->log_likelihood);
    if (!rc) rc = check_mixture(prm->log_prior);
    if (!rc) rc = check_mixture(prm->log_q);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    PcnDev pd;
    memset(&pd, 0, sizeof(pd));
    pd.bmtab = ctx->d_bmtab;
    pd.d = prm->d;
    pd.d_noise = d_noise;
    pd.beta = prm->beta;
    pd.mu = prm->mu_dev;
    pd.L = prm->L_dev;
    pd.Linv = prm->Linv_dev;
    pd.ll = to_dev(prm->log_likelihood);
    pd.lp = to_dev(prm->log_prior);
    pd.lq = to_dev(prm->log_q);
    pd.seed = prm->seed;
    pd.gid0 = prm->gid0;
    pd.noise = prm->noise;
    pd.nu = prm->nu;
    pd.mode = PCN_X_STEP;
    ASMC_REQUIRE(pd.noise == ASMC_NOISE_F64 || pd.noise == ASMC_NOISE_F32, "bad noise mode");
    // device step-size cell + history
    double* d_rho = ctx->d_rho;            // [0]: current rho
    double* d_rho_hist = ctx->d_rho + 8;   // [n_steps]
    long long* d_counts = ctx->d_counts;   // [n_steps]
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
    const bool dbg = getenv("ASMC_DEBUG_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    ASMC_HIP(hipStreamSynchronize(st));
    const double t_b = now();
    ctx->h_pinned[0] = *rho_inout_host;
    ASMC_HIP(hipMemcpyAsync(d_rho, ctx->h_pinned, sizeof(double), hipMemcpyHostToDevice, st));
    if (asmc_pcn_mm_supported(pd.d, x) && !getenv("ASMC_PCN_GENERIC")) {
        // d = 64 / 128: triangular mat-vecs on the fp64 matrix cores, always on the whitened state
        rc = asmc_pcn_mm_pack(ctx, pd, st);
        if (rc) return rc;
        int grid = 0;
        rc = asmc_pcn_mm_launch(ctx, n, prm->x_dtype, x, ll, lp, lq, pd, MM_WHITEN, d_rho, 0, d_block, &grid, st);
        if (rc) return rc;
        for (int t = 0; t < n_steps; t++) {
            rc = pcn_prepare_gamma(ctx, n, pd, step0 + (uint32_t)t, st);
            if (rc) return rc;
            rc = asmc_pcn_mm_launch(ctx, n, prm->x_dtype, x, ll, lp, lq, pd, pd.nu > 0.0 ? MM_STEP_T : MM_STEP, d_rho,
                                    step0 + (uint32_t)t, d_block, &grid, st);
            if (rc) return rc;
            rc = pcn_close_step(ctx, st, grid, d_block, n, t, d_counts, d_rho, d_rho_hist, prm->target_accept, prm->adapt, t == n_steps - 1);
            if (rc) return rc;
        }
        rc = asmc_pcn_mm_launch(ctx, n, prm->x_dtype, x, ll, lp, lq, pd, MM_UNWHITEN, d_rho, 0, d_block, &grid, st);
        if (rc) return rc;
        long long* h_counts = reinterpret_cast<long long*>(ctx->h_pinned);
        double* h_rho_hist = ctx->h_pinned + ASMC_MAX_PCN_STEPS + 8;
        ASMC_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(long long) * n_steps, hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipMemcpyAsync(h_rho_hist, d_rho_hist, sizeof(double) * n_steps, hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipMemcpyAsync(ctx->h_pinned + 8000, d_rho, sizeof(double), hipMemcpyDeviceToHost, st));
        rc = pcn_enqueue_lq_check(ctx, n, lq, st);
        if (rc) return rc;
        ASMC_HIP(hipStreamSynchronize(st));
        for (int t = 0; t < n_steps; t++) n_accept_host[t] = (int64_t)h_counts[t];
        if (rho_hist_host)
            for (int t = 0; t < n_steps; t++) rho_hist_host[t] = h_rho_hist[t];
        *rho_inout_host = ctx->h_pinned[8000];
        memcpy(&ctx->lq_nan, ctx->h_pinned + 8002, sizeof(unsigned long long));
        return ASMC_OK;
    }
    const bool reg_ok = pcn_reg_supported(pd.d, prm->x_dtype == ASMC_F64 ? 8 : 4, x);
    // whitened-state stepping: plain (single-component) targets, enough steps to amortise the two conversions
    const bool single = pd.ll.C == 1 && pd.lp.C == 1 && pd.lq.C == 1;
    bool y_state = reg_ok && n_steps >= 4 && !getenv("ASMC_PCN_XSTATE");  // several components: only coordinate-major (below)
    // whitened state in a coordinate-major scratch buffer (grown on demand, kept for the life of the ctx)
    const bool soa = y_state && pcn_ensure_ysoa(ctx, n, pd.d, prm->x_dtype, pd, st);
    if (!single && !soa) y_state = false;  // the row-major whitened-state kernel folds single Gaussians only
    auto launch_mode = [&](int mode, uint32_t stp, int* grid) -> int {
        if (soa) mode = mode == PCN_WHITEN ? PCN_WHITEN_S : mode == PCN_UNWHITEN ? PCN_UNWHITEN_S
                        : mode == PCN_Y_STEP ? (single ? PCN_Y_STEP_S : PCN_Y_STEP_SG)
                        : mode == PCN_Y_STEP_T ? (single ? PCN_Y_STEP_TS : PCN_Y_STEP_TSG) : mode;
        pd.mode = mode;
        const int nz = pd.noise;
        if (mode == PCN_WHITEN || mode == PCN_UNWHITEN || mode == PCN_UNWHITEN_X || mode == PCN_WHITEN_S || mode == PCN_UNWHITEN_S)
            pd.noise = ASMC_NOISE_F64;
        int r;
        if (prm->x_dtype == ASMC_F64)
            r = launch_pcn_step<double, 0>(ctx, n, (double*)x, ll, lp, lq, pd, d_rho, stp, d_block, grid, nullptr, nullptr, nullptr, st);
        else
            r = launch_pcn_step<float, 0>(ctx, n, (float*)x, ll, lp, lq, pd, d_rho, stp, d_block, grid, nullptr, nullptr, nullptr, st);
        pd.noise = nz;
        return r;
    };
    if (reg_ok) {
        rc = pack_pcn_tables(ctx, pd, st);
        if (rc) return rc;
    }
    int grid = 0;
    if (y_state) {
        rc = launch_mode(PCN_WHITEN, 0, &grid);
        if (rc) return rc;
    }
    for (int t = 0; t < n_steps; t++) {
        const bool tp = pd.nu > 0.0 && reg_ok;  // the generic kernel branches on the variates' pointer at run time
        rc = pcn_prepare_gamma(ctx, n, pd, step0 + (uint32_t)t, st);
        if (rc) return rc;
        rc = launch_mode(y_state ? (tp ? PCN_Y_STEP_T : PCN_Y_STEP) : (tp ? PCN_X_STEP_T : PCN_X_STEP), step0 + (uint32_t)t, &grid);
        if (rc) return rc;
        rc = pcn_close_step(ctx, st, grid, d_block, n, t, d_counts, d_rho, d_rho_hist, prm->target_accept, prm->adapt, t == n_steps - 1);
        if (rc) return rc;
    }
    if (y_state) {
        rc = launch_mode(PCN_UNWHITEN, 0, &grid);
        if (rc) return rc;
    }
    const double t_c = now();
    // read back: counts [n_steps] | rho_hist [n_steps] | rho
    long long* h_counts = reinterpret_cast<long long*>(ctx->h_pinned);
    double* h_rho_hist = ctx->h_pinned + ASMC_MAX_PCN_STEPS + 8;  // disjoint from the counts
    ASMC_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(long long) * n_steps, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(h_rho_hist, d_rho_hist, sizeof(double) * n_steps, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned + 8000, d_rho, sizeof(double), hipMemcpyDeviceToHost, st));
    rc = pcn_enqueue_lq_check(ctx, n, lq, st);
    if (rc) return rc;
    ASMC_HIP(hipStreamSynchronize(st));
    memcpy(&ctx->lq_nan, ctx->h_pinned + 8002, sizeof(unsigned long long));
    if (dbg) fprintf(stderr, "[asmc_pcn_mutate] sync0 %.3f ms, enqueue %.3f ms, drain %.3f ms (n_steps=%d)\n", t_b - t_a, t_c - t_b, now() - t_c, n_steps);
    for (int t = 0; t < n_steps; t++) n_accept_host[t] = (int64_t)h_counts[t];
    if (rho_hist_host)
        for (int t = 0; t < n_steps; t++) rho_hist_host[t] = h_rho_hist[t];
    *rho_inout_host = ctx->h_pinned[8000];
    return ASMC_OK;
}

// proposal half of the split path for the step `step` (gamma variates drawn first when pd.nu > 0): register-resident
// kernel for d in {4, 8, 16, 32}, the same kernel on identity-padded tables for 16 < d < 32, generic LDS kernel otherwise
static int pcn_propose_launch(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, void* x_prop, double* qf_old,
                              double* qf_new, PcnDev pd, const double* rho_ptr, uint32_t step, hipStream_t st) {
    int grid = 0;
    int rc = pcn_prepare_gamma(ctx, n, pd, step, st);
    if (rc) return rc;
    const bool pad = d > 16 && d < 32 && !getenv("ASMC_PCN_GENERIC");
    if (pad || (pcn_reg_supported(d, x_dtype == ASMC_F64 ? 8 : 4, x) && ((uintptr_t)x_prop % 16) == 0)) {
        // d in {4, 8, 16, 32}: the register-resident kernel's proposal half (same arithmetic as the fused step);
        // 16 < d < 32: the same kernel compiled for 32 dimensions with identity-padded tables
        pd.dpad = pad ? 32 : 0;
        pd.mode = pad ? (pd.nu > 0.0 ? PCN_X_PROPOSE_PAD_T : PCN_X_PROPOSE_PAD) : (pd.nu > 0.0 ? PCN_X_PROPOSE_T : PCN_X_PROPOSE);
        pd.noise = ASMC_NOISE_F64;
        pd.ys = x_prop;
        rc = pack_pcn_tables(ctx, pd, st);
        if (rc) return rc;
        if (x_dtype == ASMC_F64)
            return launch_pcn_step<double, 0>(ctx, n, (double*)const_cast<void*>(x), qf_old, qf_new, nullptr, pd, rho_ptr,
                                              step, nullptr, &grid, nullptr, nullptr, nullptr, st);
        return launch_pcn_step<float, 0>(ctx, n, (float*)const_cast<void*>(x), qf_old, qf_new, nullptr, pd, rho_ptr, step,
                                         nullptr, &grid, nullptr, nullptr, nullptr, st);
    }
    if (asmc_pcn_mm_supported(d, x) && ((uintptr_t)x_prop % 16) == 0 && ctx->d_mmtab && !getenv("ASMC_PCN_GENERIC")) {
        // d = 64 / 128: both mat-vecs of the proposal on the fp64 matrix cores (the generic LDS kernel took 10 / 76 ms per
        // step at 1M particles - the path every user with Python densities and d > 32 was on)
        pd.noise = ASMC_NOISE_F64;
        rc = asmc_pcn_mm_pack(ctx, pd, st);
        if (rc) return rc;
        return asmc_pcn_mm_launch(ctx, n, x_dtype, const_cast<void*>(x), qf_old, qf_new, reinterpret_cast<double*>(x_prop), pd,
                                  pd.nu > 0.0 ? MM_XPROPOSE_T : MM_XPROPOSE, rho_ptr, step, nullptr, &grid, st);
    }
    const int D = pcn_pad_dim(d);
    if (D != d && D > 0 && D <= ctx->d_max_pad && !(D == 32 && d > 16) && !getenv("ASMC_PCN_GENERIC") && !getenv("ASMC_PCN_NOPAD")) {
        // any other d <= 128 (17 .. 31 have the in-kernel padding above): zero-padded copies of the rows and of the reference's
        // tables, the D-dimensional proposal kernel with the noise masked beyond d, x' copied back without the padding
        const size_t es = x_dtype == ASMC_F64 ? 8 : 4;
        const size_t tab_doubles = (size_t)D + 2 * (size_t)D * D;
        const size_t tab_bytes = ((tab_doubles * 8 + 255) / 256) * 256;
        const size_t row_bytes = (((size_t)n * D * es + 255) / 256) * 256;
        const size_t need = tab_bytes + 2 * row_bytes;
        if (need > ctx->xpad_bytes) {
            ASMC_HIP(hipStreamSynchronize(st));
            if (ctx->d_xpad) (void)hipFree(ctx->d_xpad);
            ctx->d_xpad = nullptr;
            ctx->xpad_bytes = 0;
            if (hipMalloc(&ctx->d_xpad, need) != hipSuccess) {
                (void)hipGetLastError();
                asmc_set_error("pcn: no device memory for the zero-padded copies of a proposal (%zu bytes)", need);
                return ASMC_ERR_NOMEM;
            }
            ctx->xpad_bytes = need;
        }
        double* tab = reinterpret_cast<double*>(ctx->d_xpad);
        void* xp = reinterpret_cast<char*>(ctx->d_xpad) + tab_bytes;
        void* xq = reinterpret_cast<char*>(xp) + row_bytes;
        PcnDev src = pd;
        src.ll.C = src.lp.C = src.lq.C = 0;
        ASMC_LAUNCH(ctx, st, "k_pad_tables", k_pad_tables, dim3(1), dim3(256), 0, st, d, D, src, tab);
        ASMC_LAUNCH_CHECK();
        const int pg = grid_for(n * D, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
        if (es == 8)
            ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<double>, dim3(pg), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)x, (double*)xp);
        else
            ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<float>, dim3(pg), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)x, (float*)xp);
        ASMC_LAUNCH_CHECK();
        PcnDev p2 = pd;
        p2.d = D;
        p2.d_noise = d;
        p2.mu = tab;
        p2.L = tab + D;
        p2.Linv = p2.L + (size_t)D * D;
        rc = pcn_propose_launch(ctx, n, D, x_dtype, xp, xq, qf_old, qf_new, p2, rho_ptr, step, st);
        if (rc) return rc;
        if (es == 8)
            ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<double>, dim3(pg), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)xq, (double*)x_prop);
        else
            ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<float>, dim3(pg), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)xq, (float*)x_prop);
        ASMC_LAUNCH_CHECK();
        return ASMC_OK;
    }
    if (x_dtype == ASMC_F64)
        return launch_pcn_step<double, 1>(ctx, n, (double*)const_cast<void*>(x), nullptr, nullptr, nullptr, pd, rho_ptr,
                                          step, nullptr, &grid, (double*)x_prop, qf_old, qf_new, st);
    return launch_pcn_step<float, 1>(ctx, n, (float*)const_cast<void*>(x), nullptr, nullptr, nullptr, pd, rho_ptr, step,
                                     nullptr, &grid, (float*)x_prop, qf_old, qf_new, st);
}

int asmc_pcn_propose(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, void* x_prop, double* qf_old,
                     double* qf_new, const double* mu, const double* L, const double* Linv, double rho, double nu,
                     uint64_t seed, uint64_t gid0, uint32_t step, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && x_prop && qf_old && qf_new && mu && L && Linv, "null pointer");
    ASMC_REQUIRE(!(nu > 0.0) || nu >= 1.0, "nu must be >= 1 (or <= 0 for the Gaussian reference)");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max && d > 0 && d <= ASMC_MAX_DIMS, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(rho >= 0.0 && rho <= 1.0, "rho must be in (0, 1], or 0 for the device-resident step size");
    hipStream_t st = as_stream(stream);
    PcnDev pd;
    memset(&pd, 0, sizeof(pd));
    pd.bmtab = ctx->d_bmtab;
    pd.d = d;
    pd.mu = mu;
    pd.L = L;
    pd.Linv = Linv;
    pd.seed = seed;
    pd.gid0 = gid0;
    pd.nu = nu;
    if (rho > 0.0) {  // by kernel argument: no pinned staging, no synchronisation
        ASMC_LAUNCH(ctx, st, "k_set_scalar", k_set_scalar, dim3(1), dim3(1), 0, st, ctx->d_rho, rho);
        ASMC_LAUNCH_CHECK();
    }
    return pcn_propose_launch(ctx, n, d, x_dtype, x, x_prop, qf_old, qf_new, pd, ctx->d_rho, step, st);
}

// Split path without a host round trip per step (asmc.h): the step size lives in the ctx, asmc_pcn_accept leaves its
// count on the device, asmc_pcn_split_adapt closes the step like the fused loops do (exchange hook included).
int asmc_pcn_split_begin(asmc_ctx* ctx, double rho0, asmc_stream stream) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(rho0 > 0.0 && rho0 <= 1.0, "rho must be in (0, 1]");
    hipStream_t st = as_stream(stream);
    ASMC_LAUNCH(ctx, st, "k_set_scalar", k_set_scalar, dim3(1), dim3(1), 0, st, ctx->d_rho, rho0);
    ASMC_LAUNCH_CHECK();
    return ASMC_OK;
}

int asmc_pcn_split_adapt(asmc_ctx* ctx, int64_t n_global, double target_accept, int t, int adapt, asmc_stream stream) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(n_global > 0 && t >= 0 && t < ASMC_MAX_PCN_STEPS, "bad n_global / step index");
    hipStream_t st = as_stream(stream);
    // the accept kernel's count cell (d_keys[0]) plays the part of a one-block partial
    return pcn_close_step(ctx, st, 1, reinterpret_cast<const long long*>(ctx->d_keys), n_global, t, ctx->d_counts, ctx->d_rho,
                          ctx->d_rho + 8, target_accept, adapt);
}

int asmc_pcn_split_end(asmc_ctx* ctx, int n_steps, int64_t* n_accept_host, double* rho_hist_host, double* rho_host,
                       asmc_stream stream) {
    ASMC_REQUIRE(ctx && n_accept_host && rho_host, "null pointer");
    ASMC_REQUIRE(n_steps >= 1 && n_steps <= ASMC_MAX_PCN_STEPS, "n_steps out of range");
    hipStream_t st = as_stream(stream);
    long long* h_counts = reinterpret_cast<long long*>(ctx->h_pinned);
    double* h_rho_hist = ctx->h_pinned + ASMC_MAX_PCN_STEPS + 8;
    ASMC_HIP(hipStreamSynchronize(st));  // pinned staging may still be in flight
    ASMC_HIP(hipMemcpyAsync(h_counts, ctx->d_counts, sizeof(long long) * n_steps, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(h_rho_hist, ctx->d_rho + 8, sizeof(double) * n_steps, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned + 8000, ctx->d_rho, sizeof(double), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < n_steps; t++) n_accept_host[t] = (int64_t)h_counts[t];
    if (rho_hist_host)
        for (int t = 0; t < n_steps; t++) rho_hist_host[t] = h_rho_hist[t];
    *rho_host = ctx->h_pinned[8000];
    return ASMC_OK;
}

// ---- whitened-state session of the split path (asmc.h) -----------------------------------------------------------------
static int ysplit_pd(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* prm, PcnDev& pd) {
    ASMC_REQUIRE(ctx && prm, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference pointer");
    ASMC_REQUIRE(!(prm->nu > 0.0) || prm->nu >= 1.0, "nu must be >= 1 (or <= 0 for the Gaussian reference)");
    memset(&pd, 0, sizeof(pd));
    pd.bmtab = ctx->d_bmtab;
    pd.d = prm->d;
    pd.beta = prm->beta;
    pd.mu = prm->mu_dev;
    pd.L = prm->L_dev;
    pd.Linv = prm->Linv_dev;
    pd.seed = prm->seed;
    pd.gid0 = prm->gid0;
    pd.nu = prm->nu;
    ASMC_REQUIRE(prm->noise == ASMC_NOISE_F64 || prm->noise == ASMC_NOISE_F32, "bad noise mode");
    pd.noise = prm->noise;  // same in every call of a session: accept regenerates what propose drew
    return ASMC_OK;
}

int asmc_pcn_ysplit_begin(asmc_ctx* ctx, int64_t n, const void* x, const asmc_pcn_params* prm, double rho0,
                          asmc_stream stream) {
    PcnDev pd;
    int rc = ysplit_pd(ctx, n, prm, pd);
    if (rc) return rc;
    ASMC_REQUIRE(x != nullptr, "null pointer");
    ASMC_REQUIRE(rho0 > 0.0 && rho0 <= 1.0, "rho must be in (0, 1]");
    hipStream_t st = as_stream(stream);
    if (!pcn_reg_supported(pd.d, prm->x_dtype == ASMC_F64 ? 8 : 4, x) || !pcn_ensure_ysoa(ctx, n, pd.d, prm->x_dtype, pd, st)) {
        asmc_set_error("whitened-state split session: d=%d is not a register-kernel dimension or no scratch", pd.d);
        return ASMC_ERR_UNSUPPORTED;
    }
    ASMC_LAUNCH(ctx, st, "k_set_scalar", k_set_scalar, dim3(1), dim3(1), 0, st, ctx->d_rho, rho0);
    ASMC_LAUNCH_CHECK();
    rc = pack_pcn_tables(ctx, pd, st);
    if (rc) return rc;
    ctx->ptab_tag = ++ctx->ysplit_seq;
    pd.mode = PCN_WHITEN_S;
    pd.noise = ASMC_NOISE_F64;  // (the whitening kernels draw nothing: they are instantiated under this key only)
    int grid = 0;
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
    if (prm->x_dtype == ASMC_F64)
        return launch_pcn_step<double, 0>(ctx, n, (double*)const_cast<void*>(x), nullptr, nullptr, nullptr, pd, ctx->d_rho, 0, d_block,
                                          &grid, nullptr, nullptr, nullptr, st);
    return launch_pcn_step<float, 0>(ctx, n, (float*)const_cast<void*>(x), nullptr, nullptr, nullptr, pd, ctx->d_rho, 0, d_block, &grid,
                                     nullptr, nullptr, nullptr, st);
}

int asmc_pcn_ysplit_propose(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* prm, uint32_t step, void* x_prop,
                            asmc_stream stream) {
    PcnDev pd;
    int rc = ysplit_pd(ctx, n, prm, pd);
    if (rc) return rc;
    ASMC_REQUIRE(x_prop != nullptr && ((uintptr_t)x_prop % 16) == 0, "x_prop must be a 16-byte aligned device buffer");
    ASMC_REQUIRE(ctx->d_ysoa != nullptr, "no session (asmc_pcn_ysplit_begin)");
    hipStream_t st = as_stream(stream);
    pd.ys = ctx->d_ysoa;
    pd.n_pad = ((n + 63) / 64) * 64;
    if (ctx->ptab_tag == 0 || ctx->ptab_tag != ctx->ysplit_seq) {
        // someone else packed parameter tables on this context since asmc_pcn_ysplit_begin (cheap insurance; a context
        // carries ONE mutation at a time - step size, counts and scale variates are per context - see include/asmc.h)
        rc = pack_pcn_tables(ctx, pd, st);
        if (rc) return rc;
        ctx->ptab_tag = ctx->ysplit_seq;
    }
    rc = pcn_prepare_gamma(ctx, n, pd, step, st);
    if (rc) return rc;
    int grid = 0;
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
    if (prm->x_dtype == ASMC_F64)
        return dispatch_pcn_reg_flow<double, PCN_FLOW_PROPOSE_SX>(ctx, n, nullptr, (double*)x_prop, nullptr, nullptr, nullptr, nullptr,
                                                                  nullptr, nullptr, pd, ctx->d_rho, step, d_block, &grid, st);
    return dispatch_pcn_reg_flow<float, PCN_FLOW_PROPOSE_SX>(ctx, n, nullptr, (float*)x_prop, nullptr, nullptr, nullptr, nullptr,
                                                             nullptr, nullptr, pd, ctx->d_rho, step, d_block, &grid, st);
}

// packs the transform epilogue's table (TRTAB_*) from the device-resident transform / premap / mixture tables
__global__ void k_trtab_pack(int d, const int* __restrict__ kind, const double* __restrict__ lower, const double* __restrict__ upper,
                             const double* __restrict__ mean, const double* __restrict__ sd, const double* __restrict__ premap,
                             const double* __restrict__ q_logw, const double* __restrict__ q_mu, const double* __restrict__ q_prec,
                             double eps, double unit_logj, double affine_logj, int minus_logj, double* __restrict__ out) {
    const int j = threadIdx.x;
    if (j < d) {
        out[j] = (double)kind[j];
        out[d + j] = lower[j];
        out[2 * d + j] = upper[j];
        out[3 * d + j] = mean ? mean[j] : 0.0;
        out[4 * d + j] = sd ? sd[j] : 1.0;
        out[5 * d + j] = 1.0 / (upper[j] - lower[j]);
        out[6 * d + j] = sd ? 1.0 / sd[j] : 1.0;
        for (int r = 0; r < 5; r++) out[(7 + r) * d + j] = premap ? premap[r * d + j] : 0.0;
        out[12 * d + j] = q_mu ? q_mu[j] : 0.0;
        out[13 * d + j] = q_prec ? q_prec[j] : 0.0;
    }
    if (j == 0) {
        double* sc = out + TRTAB_ROWS * d;
        sc[0] = eps, sc[1] = unit_logj, sc[2] = affine_logj, sc[3] = mean ? 1.0 : 0.0;
        sc[4] = q_logw ? q_logw[0] : 0.0, sc[5] = premap ? 1.0 : 0.0, sc[6] = minus_logj ? 1.0 : 0.0, sc[7] = 0.0;
    }
}

int asmc_pcn_ysplit_propose_tr(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* prm, uint32_t step, const asmc_transform* t,
                               const double* premap_dev, const asmc_mixture* qmix, int minus_logj, void* x_prop,
                               double* logj_out, double* lq_out, asmc_stream stream) {
    PcnDev pd;
    int rc = ysplit_pd(ctx, n, prm, pd);
    if (rc) return rc;
    ASMC_REQUIRE(t && x_prop && logj_out && ((uintptr_t)x_prop % 16) == 0, "null / misaligned pointer");
    ASMC_REQUIRE(ctx->d_ysoa != nullptr, "no session (asmc_pcn_ysplit_begin)");
    ASMC_REQUIRE(t->d == pd.d && t->kind_dev && t->lower_dev && t->upper_dev, "transform tables missing or of another dimension");
    ASMC_REQUIRE((t->mean_dev == nullptr) == (t->std_dev == nullptr), "affine stage needs both mean and std");
    ASMC_REQUIRE((premap_dev == nullptr) == (qmix == nullptr) && (premap_dev == nullptr) == (lq_out == nullptr),
                 "premap, mixture and lq_out come together");
    ASMC_REQUIRE(!qmix || qmix->n_components == 1, "the proposal density must be a single Gaussian");
    const int logit = (t->hints & 7) == (ASMC_TR_NO_PERIODIC | ASMC_TR_NO_PROBIT);
    const int probit = (t->hints & 7) == (ASMC_TR_NO_PERIODIC | ASMC_TR_NO_LOGIT);
    if (!logit && !probit) {
        asmc_set_error("asmc_pcn_ysplit_propose_tr: the transform must be non-periodic with ONE bounded stage (hints %d)", t->hints);
        return ASMC_ERR_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    pd.ys = ctx->d_ysoa;
    pd.n_pad = ((n + 63) / 64) * 64;
    if (ctx->ptab_tag == 0 || ctx->ptab_tag != ctx->ysplit_seq) {
        rc = pack_pcn_tables(ctx, pd, st);
        if (rc) return rc;
        ctx->ptab_tag = ctx->ysplit_seq;
    }
    rc = pcn_prepare_gamma(ctx, n, pd, step, st);
    if (rc) return rc;
    double* d_tab = ctx->d_small + 3072;  // TRTAB_ROWS * 32 + TRTAB_SCALARS doubles
    ASMC_LAUNCH(ctx, st, "k_trtab_pack", k_trtab_pack, dim3(1), dim3(64), 0, st, pd.d, t->kind_dev, t->lower_dev, t->upper_dev,
                t->mean_dev, t->std_dev, premap_dev, qmix ? qmix->logw_dev : (const double*)nullptr,
                qmix ? qmix->mu_dev : (const double*)nullptr, qmix ? qmix->prec_dev : (const double*)nullptr, t->eps, t->unit_logj,
                t->affine_logj, minus_logj, d_tab);
    ASMC_LAUNCH_CHECK();
    int grid = 0;
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
#define PROPOSE_TR(TT, MD) \
    dispatch_pcn_reg_flow<TT, MD>(ctx, n, nullptr, (TT*)x_prop, nullptr, nullptr, nullptr, logj_out, lq_out, (const double*)d_tab, pd, \
                                  ctx->d_rho, step, d_block, &grid, st)
    if (prm->x_dtype == ASMC_F64) return logit ? PROPOSE_TR(double, PCN_FLOW_PROPOSE_SXT_LOGIT) : PROPOSE_TR(double, PCN_FLOW_PROPOSE_SXT_PROBIT);
    return logit ? PROPOSE_TR(float, PCN_FLOW_PROPOSE_SXT_LOGIT) : PROPOSE_TR(float, PCN_FLOW_PROPOSE_SXT_PROBIT);
#undef PROPOSE_TR
}

int asmc_pcn_ysplit_accept(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* prm, uint32_t step, double* ll, double* lp,
                           double* lq, const double* ll_new, const double* lp_new, const double* lq_new, double* lj,
                           const double* lj_new, int64_t n_global, int t, asmc_stream stream) {
    PcnDev pd;
    int rc = ysplit_pd(ctx, n, prm, pd);
    if (rc) return rc;
    ASMC_REQUIRE(ll && lp && lq && ll_new && lp_new && lq_new, "null pointer");
    ASMC_REQUIRE((lj == nullptr) == (lj_new == nullptr), "log-Jacobian arrays: both or neither");
    ASMC_REQUIRE(ctx->d_ysoa != nullptr, "no session (asmc_pcn_ysplit_begin)");
    ASMC_REQUIRE(n_global > 0 && t >= 0 && t < ASMC_MAX_PCN_STEPS, "bad n_global / step index");
    hipStream_t st = as_stream(stream);
    pd.ys = ctx->d_ysoa;
    pd.n_pad = ((n + 63) / 64) * 64;
    rc = pcn_prepare_gamma(ctx, n, pd, step, st);  // the variates asmc_pcn_ysplit_propose drew for this step (still there; else redrawn: counter based)
    if (rc) return rc;
    int grid = 0;
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
    if (lj && prm->x_dtype == ASMC_F64)
        rc = dispatch_pcn_reg_flow<double, PCN_FLOW_ACCEPT_SJ>(ctx, n, lj, const_cast<double*>(lj_new), ll, lp, lq,
                                                               const_cast<double*>(ll_new), const_cast<double*>(lp_new), lq_new,
                                                               pd, ctx->d_rho, step, d_block, &grid, st);
    else if (lj)
        rc = dispatch_pcn_reg_flow<float, PCN_FLOW_ACCEPT_SJ>(
            ctx, n, static_cast<float*>(static_cast<void*>(lj)), static_cast<float*>(static_cast<void*>(const_cast<double*>(lj_new))), ll,
            lp, lq, const_cast<double*>(ll_new), const_cast<double*>(lp_new), lq_new, pd, ctx->d_rho, step, d_block, &grid, st);
    else if (prm->x_dtype == ASMC_F64)
        rc = dispatch_pcn_reg_flow<double, PCN_FLOW_ACCEPT_S>(ctx, n, nullptr, nullptr, ll, lp, lq, const_cast<double*>(ll_new),
                                                              const_cast<double*>(lp_new), lq_new, pd, ctx->d_rho, step, d_block,
                                                              &grid, st);
    else
        rc = dispatch_pcn_reg_flow<float, PCN_FLOW_ACCEPT_S>(ctx, n, nullptr, nullptr, ll, lp, lq, const_cast<double*>(ll_new),
                                                             const_cast<double*>(lp_new), lq_new, pd, ctx->d_rho, step, d_block,
                                                             &grid, st);
    if (rc) return rc;
    return pcn_close_step(ctx, st, grid, d_block, n_global, t, ctx->d_counts, ctx->d_rho, ctx->d_rho + 8, prm->target_accept,
                          prm->adapt);
}

int asmc_pcn_ysplit_end(asmc_ctx* ctx, int64_t n, void* x, const asmc_pcn_params* prm, asmc_stream stream) {
    PcnDev pd;
    int rc = ysplit_pd(ctx, n, prm, pd);
    if (rc) return rc;
    ASMC_REQUIRE(x != nullptr && ctx->d_ysoa != nullptr, "null pointer / no session");
    hipStream_t st = as_stream(stream);
    pd.ys = ctx->d_ysoa;
    pd.n_pad = ((n + 63) / 64) * 64;
    pd.mode = PCN_UNWHITEN_XS;
    pd.noise = ASMC_NOISE_F64;
    rc = pack_pcn_tables(ctx, pd, st);
    if (rc) return rc;
    int grid = 0;
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
    if (prm->x_dtype == ASMC_F64)
        return launch_pcn_step<double, 0>(ctx, n, (double*)x, nullptr, nullptr, nullptr, pd, ctx->d_rho, 0, d_block, &grid, nullptr,
                                          nullptr, nullptr, st);
    return launch_pcn_step<float, 0>(ctx, n, (float*)x, nullptr, nullptr, nullptr, pd, ctx->d_rho, 0, d_block, &grid, nullptr, nullptr,
                                     nullptr, st);
}

int asmc_pcn_accept(asmc_ctx* ctx, int64_t n, int d, int x_dtype, void* x, const void* x_prop, double* ll, double* lp,
                    double* lq, const double* ll_new, const double* lp_new, const double* lq_new,
                    double* lj_old, const double* lj_new, const double* qf_old, const double* qf_new,
                    double beta, uint64_t seed, uint64_t gid0, uint32_t step, int64_t* n_accept_host,
                    asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && x_prop && ll && lp && lq && ll_new && lp_new && lq_new && qf_old && qf_new, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max && d > 0, "bad sizes");
    ASMC_REQUIRE(x_dtype == ASMC_F64 || x_dtype == ASMC_F32, "bad x_dtype");
    hipStream_t st = as_stream(stream);
    unsigned char* flags = ctx->d_flags;
    ASMC_HIP(hipMemsetAsync(ctx->d_keys, 0, sizeof(unsigned long long), st));
    const int grid = grid_for(n, ASMC_BLOCK * 2, ASMC_MAX_BLOCKS);
    ASMC_LAUNCH(ctx, st, "k_pcn_accept_flags", k_pcn_accept_flags, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq, ll_new, lp_new, lq_new,
                       lj_old, lj_new, qf_old, qf_new, beta, (unsigned long long)seed, (unsigned long long)gid0, step,
                       flags, ctx->d_keys);
    ASMC_LAUNCH_CHECK();
    {
        const int rc2 = launch_copy_flagged(ctx, n, d, x_dtype, x, x_prop, flags, st);
        if (rc2) return rc2;
    }
    if (n_accept_host) {
        unsigned long long* h = reinterpret_cast<unsigned long long*>(ctx->h_pinned);
        ASMC_HIP(hipMemcpyAsync(h, ctx->d_keys, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        ASMC_HIP(hipStreamSynchronize(st));
        *n_accept_host = (int64_t)h[0];
    }
    return ASMC_OK;
}

// pCN with a coupling-flow proposal density: per step  propose -> flow log q(x') (fp32 MFMA, asmc_flow.hip) ->
// built-in log prior / log likelihood of x' -> accept -> device-side step-size adaptation; no host round trip
// inside the loop.  Work buffer: x' [n, d] (storage type) | q0 | q1 | ll' | lp' | lq'  (fp64 [n] each).
int64_t asmc_pcn_flow_work_bytes(int64_t n, int d, int x_dtype) {
    const int64_t xb = ((n * d * (x_dtype == ASMC_F64 ? 8 : 4) + 255) / 256) * 256;
    return xb + 5 * (((n * 8 + 255) / 256) * 256);
}

// waits for a flow-mutation call's read-back (counts | step sizes | rho | non-finite count | NaNs of log q) and hands it out
static int mutate_flow_collect(asmc_ctx* ctx, int n_steps, double* rho_out_host, int64_t* n_accept_host, double* rho_hist_host,
                               hipStream_t st, hipEvent_t ev = nullptr) {
    const long long* h_counts = reinterpret_cast<const long long*>(ctx->h_pinned);
    const double* h_rho_hist = ctx->h_pinned + ASMC_MAX_PCN_STEPS + 8;
    if (ev)
        ASMC_HIP(hipEventSynchronize(ev));
    else
        ASMC_HIP(hipStreamSynchronize(st));
    memcpy(&ctx->lq_nan, ctx->h_pinned + 8002, sizeof(unsigned long long));
    for (int t = 0; t < n_steps; t++) n_accept_host[t] = (int64_t)h_counts[t];
    if (rho_hist_host)
        for (int t = 0; t < n_steps; t++) rho_hist_host[t] = h_rho_hist[t];
    *rho_out_host = h_rho_hist[-8];
    memcpy(&ctx->flow_nonfinite, ctx->h_pinned + 8001, sizeof(unsigned long long));  // fused steps only (else stale zero)
    return ASMC_OK;
}

}  // extern "C"

// d_noise > 0: a d_noise-dimensional problem zero-padded to prm->d = 32 by asmc_pcn_mutate_flow below (the flow keeps its own
// dims = d_noise); only the one-kernel step takes it
static int pcn_mutate_flow_impl(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq,
                                const asmc_pcn_params* prm, const asmc_coupling* flow, void* work_dev, int64_t work_bytes,
                                int n_steps, uint32_t step0, double* rho_inout_host, int64_t* n_accept_host,
                                double* rho_hist_host, asmc_stream stream, int d_noise) {
    ASMC_REQUIRE(ctx && x && ll && lp && lq && prm && flow && work_dev && rho_inout_host && n_accept_host, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    ASMC_REQUIRE(n_steps >= 1 && n_steps <= ASMC_MAX_PCN_STEPS, "n_steps out of range (<= 2048 per call)");
    ASMC_REQUIRE(prm->d > 0 && prm->d <= ASMC_MAX_DIMS && (d_noise > 0 ? d_noise : prm->d) == flow->dims, "bad d");
    ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference-Gaussian pointer");
    ASMC_REQUIRE(*rho_inout_host > 0.0 && *rho_inout_host <= 1.0, "rho must be in (0, 1]");
    ASMC_REQUIRE(d_noise > 0 || work_bytes >= asmc_pcn_flow_work_bytes(n, prm->d, prm->x_dtype), "work buffer too small");
    ASMC_REQUIRE(!(prm->nu > 0.0) || prm->nu >= 1.0, "nu must be >= 1 (or <= 0 for the Gaussian reference)");
    int rc = check_mixture(prm->log_likelihood);
    if (!rc) rc = check_mixture(prm->log_prior);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const int d = prm->d;
    const int64_t xb = ((n * d * (prm->x_dtype == ASMC_F64 ? 8 : 4) + 255) / 256) * 256;
    const int64_t vb = ((n * 8 + 255) / 256) * 256;
    char* w = reinterpret_cast<char*>(work_dev);
    void* x_prop = w;
    double* q0 = reinterpret_cast<double*>(w + xb);
    double* q1 = reinterpret_cast<double*>(w + xb + vb);
    double* ll_new = reinterpret_cast<double*>(w + xb + 2 * vb);
    double* lp_new = reinterpret_cast<double*>(w + xb + 3 * vb);
    double* lq_new = reinterpret_cast<double*>(w + xb + 4 * vb);
    PcnDev pd;
    memset(&pd, 0, sizeof(pd));
    pd.bmtab = ctx->d_bmtab;
    pd.d = d;
    pd.mu = prm->mu_dev;
    pd.L = prm->L_dev;
    pd.Linv = prm->Linv_dev;
    pd.seed = prm->seed;
    pd.gid0 = prm->gid0;
    pd.nu = prm->nu;
    pd.d_noise = d_noise;
    double* d_rho = ctx->d_rho;
    double* d_rho_hist = ctx->d_rho + 8;
    long long* d_counts = ctx->d_counts;
    unsigned long long* d_cnt = ctx->d_keys;  // accept counter of the current step
    unsigned char* flags = ctx->d_flags;
    // the step size goes to the device as a kernel argument: no pinned staging, so no synchronisation in front of this call's
    // launches - the host enqueues the whole mutation while the reference fit's passes are still running
    // register-resident whitened-state path (d in {4, 8, 16, 32}): x -> y once, then per step
    // propose / flow / accept / adapt, y -> x at the end; four launches per step, no host round trip
    const bool reg_ok = pcn_reg_supported(d, prm->x_dtype == ASMC_F64 ? 8 : 4, x) && !getenv("ASMC_PCN_XSTATE");
    if (!reg_ok) {  // (the register path's table pack carries the step size)
        ASMC_LAUNCH(ctx, st, "k_set_scalar", k_set_scalar, dim3(1), dim3(1), 0, st, d_rho, *rho_inout_host);
        ASMC_LAUNCH_CHECK();
    }
    if (reg_ok) {
        pd.beta = prm->beta;
        pd.ll = to_dev(prm->log_likelihood);
        pd.lp = to_dev(prm->log_prior);
        pd.lq = pd.lp;  // placeholder: the proposal density is the flow
        pd.noise = prm->noise;
        ASMC_REQUIRE(pd.noise == ASMC_NOISE_F64 || pd.noise == ASMC_NOISE_F32, "bad noise mode");
        rc = pack_pcn_tables(ctx, pd, st, d_rho, *rho_inout_host);
        if (rc) return rc;
        long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
        int grid = 0;
        const bool soa = pcn_ensure_ysoa(ctx, n, d, prm->x_dtype, pd, st, asmc_pcn_flow_fused_ok(prm, flow));
        auto convert = [&](int mode) -> int {
            PcnDev pc = pd;
            pc.mode = !soa ? mode : mode == PCN_WHITEN ? PCN_WHITEN_S : PCN_UNWHITEN_XS;
            pc.noise = ASMC_NOISE_F64;
            if (prm->x_dtype == ASMC_F64)
                return launch_pcn_step<double, 0>(ctx, n, (double*)x, ll, lp, lq, pc, d_rho, 0, d_block, &grid, nullptr, nullptr, nullptr, st);
            return launch_pcn_step<float, 0>(ctx, n, (float*)x, ll, lp, lq, pc, d_rho, 0, d_block, &grid, nullptr, nullptr, nullptr, st);
        };
        rc = convert(PCN_WHITEN);
        if (rc) return rc;
        // one kernel per step (propose -> flow on the MFMA -> targets -> accept) where the shape allows it
        const bool fused = soa && asmc_pcn_flow_fused_ok(prm, flow);
        if (d_noise > 0 && !fused) {
            asmc_set_error("asmc_pcn_mutate_flow: the zero-padded form needs the one-kernel step");
            return ASMC_ERR_UNSUPPORTED;
        }
        // counters of the fused steps: [t] tile hand-out, [ASMC_MAX_PCN_STEPS + t] blocks done
        // (sizes in multiples of 16 bytes: a ragged memset is two fill kernels)
        if (fused) {  // ... and every tile starts in half 0 of the state allocation (tile parities: the split path's flag bytes,
            // free here, which sit right behind the counters: ONE fill)
            ASMC_HIP(hipMemsetAsync(ctx->d_tilectr, 0, ASMC_TILECTR_BYTES + ((size_t)((n + 63) / 64) + 15) / 16 * 16, st));
            pd.tile_par = ctx->d_flags;
        }
        for (int t = 0; t < (fused ? n_steps : 0); t++) {
            const uint32_t step = step0 + (uint32_t)t;
            rc = pcn_prepare_gamma(ctx, n, pd, step, st);
            if (rc) return rc;
            // single rank: the step's last block adapts the step size itself; sharded: the ranks' counts are exchanged
            // between the steps and the next step's prologue adapts (PcnAdaptArgs)
            PcnAdaptArgs ad = {ctx->d_tilectr + ASMC_MAX_PCN_STEPS + t,
                               reinterpret_cast<unsigned long long*>(ctx->d_tilectr + 2 * ASMC_MAX_PCN_STEPS),
                               ctx->count_hook ? ctx->count_cell : nullptr,
                               ctx->count_hook && t > 0 ? ctx->count_cell : nullptr, d_counts, d_rho, d_rho_hist,
                               prm->target_accept, ctx->count_hook ? ctx->count_n_global : n, t, prm->adapt, t == n_steps - 1};
            const int lag = prm->adapt >= 2 ? prm->adapt : 1;
            if (ctx->count_hook && lag > ctx->count_cells) {
                asmc_set_error("adapt_lag %d needs %d exchange cells, the installed hook has %d (asmc_pcn_set_count_cells)", lag, lag, ctx->count_cells);
                return ASMC_ERR_ARG;
            }
            rc = asmc_pcn_flow_fused_launch(ctx, n, prm->x_dtype == ASMC_F64 ? ASMC_F64 : ASMC_F32, ll, lp, lq, pd, flow, d_rho, step,
                                            ctx->d_tilectr + t, d_block, &grid, ad, st);
            if (rc) return rc;
            // the kernel's last block left this rank's count in the step's cell: exchange it - lagged runs at the end of a block
            if (ctx->count_hook && ((t + 1) % lag == 0 || t == n_steps - 1)) {
                const int hrc = ctx->count_hook(ctx->count_hook_user, reinterpret_cast<asmc_stream>(st));
                if (hrc != 0) {
                    asmc_set_error("accept-count exchange hook failed (%d)", hrc);
                    return ASMC_ERR_ARG;
                }
                if (t == n_steps - 1) {  // nobody's prologue follows the last step
                    ASMC_LAUNCH(ctx, st, "k_pcn_adapt", k_pcn_adapt, dim3(1), dim3(1024), 0, st, 1,
                                (const long long*)ctx->count_cell, ctx->count_n_global, t, d_counts, d_rho, d_rho_hist,
                                prm->target_accept, prm->adapt, (const double*)(d_rho_hist + t), 1,
                                lag > 1 ? (const long long*)ctx->count_cell : (const long long*)nullptr);
                    ASMC_LAUNCH_CHECK();
                }
            }
        }
        for (int t = 0; t < (fused ? 0 : n_steps); t++) {
            const uint32_t step = step0 + (uint32_t)t;
            rc = pcn_prepare_gamma(ctx, n, pd, step, st);
            if (rc) return rc;
#define FLOW_STEP(TT, MD)                                                                                               \
    dispatch_pcn_reg_flow<TT, MD>(ctx, n, (TT*)x, (TT*)x_prop, ll, lp, lq, ll_new, lp_new, lq_new, pd, d_rho, step, d_block, \
                                  &grid, st)
            if (prm->x_dtype == ASMC_F64)
                rc = soa ? FLOW_STEP(double, PCN_FLOW_PROPOSE_S) : FLOW_STEP(double, PCN_FLOW_PROPOSE);
            else
                rc = soa ? FLOW_STEP(float, PCN_FLOW_PROPOSE_S) : FLOW_STEP(float, PCN_FLOW_PROPOSE);
            if (rc) return rc;
            rc = asmc_coupling_logprob(ctx, n, prm->x_dtype, x_prop, flow, lq_new, stream);
            if (rc) return rc;
            if (prm->x_dtype == ASMC_F64)
                rc = soa ? FLOW_STEP(double, PCN_FLOW_ACCEPT_S) : FLOW_STEP(double, PCN_FLOW_ACCEPT);
            else
                rc = soa ? FLOW_STEP(float, PCN_FLOW_ACCEPT_S) : FLOW_STEP(float, PCN_FLOW_ACCEPT);
#undef FLOW_STEP
            if (rc) return rc;
            rc = pcn_close_step(ctx, st, grid, d_block, n, t, d_counts, d_rho, d_rho_hist, prm->target_accept, prm->adapt, t == n_steps - 1);
            if (rc) return rc;
        }
        rc = convert(PCN_UNWHITEN_X);
        if (rc) return rc;
    }
    if (d_noise > 0 && !reg_ok) {
        asmc_set_error("asmc_pcn_mutate_flow: the zero-padded form needs the one-kernel step");
        return ASMC_ERR_UNSUPPORTED;
    }
    for (int t = 0; t < (reg_ok ? 0 : n_steps); t++) {
        const uint32_t step = step0 + (uint32_t)t;
        rc = pcn_propose_launch(ctx, n, d, prm->x_dtype, x, x_prop, q0, q1, pd, d_rho, step, st);
        if (rc) return rc;
        rc = asmc_coupling_logprob(ctx, n, prm->x_dtype, x_prop, flow, lq_new, stream);
        if (rc) return rc;
        rc = asmc_mixture_logpdf(ctx, n, d, prm->x_dtype, x_prop, &prm->log_prior, lp_new, stream);
        if (rc) return rc;
        rc = asmc_mixture_logpdf(ctx, n, d, prm->x_dtype, x_prop, &prm->log_likelihood, ll_new, stream);
        if (rc) return rc;
        ASMC_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), st));
        const int g1 = grid_for(n, ASMC_BLOCK * 2, ASMC_MAX_BLOCKS);
        ASMC_LAUNCH(ctx, st, "k_pcn_accept_flags", k_pcn_accept_flags, dim3(g1), dim3(ASMC_BLOCK), 0, st, n, ll, lp, lq,
                    (const double*)ll_new, (const double*)lp_new, (const double*)lq_new, (double*)nullptr,
                    (const double*)nullptr, (const double*)q0, (const double*)q1, prm->beta, (unsigned long long)prm->seed,
                    (unsigned long long)prm->gid0, step, flags, d_cnt);
        ASMC_LAUNCH_CHECK();
        rc = launch_copy_flagged(ctx, n, d, prm->x_dtype, x, x_prop, flags, st);
        if (rc) return rc;
        rc = pcn_close_step(ctx, st, 1, (const long long*)d_cnt, n, t, d_counts, d_rho, d_rho_hist, prm->target_accept, prm->adapt, t == n_steps - 1);
        if (rc) return rc;
    }
    long long* h_counts = reinterpret_cast<long long*>(ctx->h_pinned);
    double* h_rho_hist = ctx->h_pinned + ASMC_MAX_PCN_STEPS + 8;
    ASMC_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(long long) * n_steps, hipMemcpyDeviceToHost, st));
    // the step size and its history in one copy (d_rho_hist = d_rho + 8): rho lands at h_rho_hist[-8]
    ASMC_HIP(hipMemcpyAsync(h_rho_hist - 8, d_rho, sizeof(double) * (8 + n_steps), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned + 8001, ctx->d_tilectr + 2 * ASMC_MAX_PCN_STEPS, sizeof(unsigned long long),
                            hipMemcpyDeviceToHost, st));
    rc = pcn_enqueue_lq_check(ctx, n, lq, st);
    if (rc) return rc;
    if (ctx->mutate_defer) {  // asmc_pcn_mutate_flow_enqueue: the results wait in pinned memory for asmc_pcn_mutate_flow_result
        if (!ctx->ev_mutate) ASMC_HIP(hipEventCreateWithFlags(&ctx->ev_mutate, hipEventDisableTiming));
        ASMC_HIP(hipEventRecord(ctx->ev_mutate, st));  // _result waits for THIS point, not for what the caller enqueues behind it
        ctx->mutate_pending_steps = n_steps;
        return ASMC_OK;
    }
    return mutate_flow_collect(ctx, n_steps, rho_inout_host, n_accept_host, rho_hist_host, st);
}

// 32 < d <= 128 (x: the state with D = 64 / 128 columns - the problem itself or its zero-padded copy, d_noise = its real
// dimension then): whiten, n_steps launches of the one-kernel step on 16-particle groups (asmc_flow16.hip), un-whiten.  The
// carried ll / lp / lq are the step kernel's own (MM_UNWHITEN_X re-evaluates nothing).
static int pcn_mutate_flow16_impl(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq, const asmc_pcn_params* prm,
                                  const asmc_coupling* flow, int n_steps, uint32_t step0, double* rho_inout_host,
                                  int64_t* n_accept_host, double* rho_hist_host, asmc_stream stream, int d_noise) {
    ASMC_REQUIRE(ctx && x && ll && lp && lq && prm && flow && rho_inout_host && n_accept_host, "null pointer");
    ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
    ASMC_REQUIRE(n_steps >= 1 && n_steps <= ASMC_MAX_PCN_STEPS, "n_steps out of range (<= 2048 per call)");
    ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
    ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference-Gaussian pointer");
    ASMC_REQUIRE(*rho_inout_host > 0.0 && *rho_inout_host <= 1.0, "rho must be in (0, 1]");
    ASMC_REQUIRE(!(prm->nu > 0.0) || prm->nu >= 1.0, "nu must be >= 1 (or <= 0 for the Gaussian reference)");
    ASMC_REQUIRE(prm->noise == ASMC_NOISE_F64 || prm->noise == ASMC_NOISE_F32, "bad noise mode");
    ASMC_REQUIRE(ctx->d_mmtab != nullptr && ((uintptr_t)x % 16) == 0, "flow16: ctx was created with d_max <= 32, or a misaligned state");
    hipStream_t st = as_stream(stream);
    PcnDev pd;
    memset(&pd, 0, sizeof(pd));
    pd.bmtab = ctx->d_bmtab;
    pd.d = prm->d;
    pd.d_noise = d_noise;
    pd.beta = prm->beta;
    pd.mu = prm->mu_dev;
    pd.L = prm->L_dev;
    pd.Linv = prm->Linv_dev;
    pd.ll = to_dev(prm->log_likelihood);
    pd.lp = to_dev(prm->log_prior);
    pd.lq = pd.lp;  // placeholder: the proposal density is the flow
    pd.seed = prm->seed;
    pd.gid0 = prm->gid0;
    pd.noise = prm->noise;
    pd.nu = prm->nu;
    double* d_rho = ctx->d_rho;
    double* d_rho_hist = ctx->d_rho + 8;
    long long* d_counts = ctx->d_counts;
    long long* d_block = ctx->d_counts + ASMC_MAX_PCN_STEPS;
    unsigned long long* d_bad = reinterpret_cast<unsigned long long*>(ctx->d_tilectr + 2 * ASMC_MAX_PCN_STEPS);
    ASMC_LAUNCH(ctx, st, "k_set_scalar", k_set_scalar, dim3(1), dim3(1), 0, st, d_rho, *rho_inout_host);
    ASMC_LAUNCH_CHECK();
    ASMC_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), st));
    int rc = asmc_pcn_mm_pack(ctx, pd, st);
    if (rc) return rc;
    rc = asmc_pcn_flow16_tables(ctx, pd, flow, st);
    if (rc) return rc;
    int grid = 0;
    rc = asmc_pcn_mm_launch(ctx, n, prm->x_dtype, x, ll, lp, lq, pd, MM_WHITEN, d_rho, 0, d_block, &grid, st);
    if (rc) return rc;
    for (int t = 0; t < n_steps; t++) {
        const uint32_t step = step0 + (uint32_t)t;
        rc = pcn_prepare_gamma(ctx, n, pd, step, st);
        if (rc) return rc;
        rc = asmc_pcn_flow16_launch(ctx, n, prm->x_dtype, x, ll, lp, lq, pd, flow, d_rho, step, d_block, &grid, d_bad, st);
        if (rc) return rc;
        rc = pcn_close_step(ctx, st, grid, d_block, n, t, d_counts, d_rho, d_rho_hist, prm->target_accept, prm->adapt, t == n_steps - 1);
        if (rc) return rc;
    }
    rc = asmc_pcn_mm_launch(ctx, n, prm->x_dtype, x, ll, lp, lq, pd, MM_UNWHITEN_X, d_rho, 0, d_block, &grid, st);
    if (rc) return rc;
    long long* h_counts = reinterpret_cast<long long*>(ctx->h_pinned);
    double* h_rho_hist = ctx->h_pinned + ASMC_MAX_PCN_STEPS + 8;
    ASMC_HIP(hipMemcpyAsync(h_counts, d_counts, sizeof(long long) * n_steps, hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(h_rho_hist - 8, d_rho, sizeof(double) * (8 + n_steps), hipMemcpyDeviceToHost, st));
    ASMC_HIP(hipMemcpyAsync(ctx->h_pinned + 8001, d_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    rc = pcn_enqueue_lq_check(ctx, n, lq, st);
    if (rc) return rc;
    if (ctx->mutate_defer) {
        if (!ctx->ev_mutate) ASMC_HIP(hipEventCreateWithFlags(&ctx->ev_mutate, hipEventDisableTiming));
        ASMC_HIP(hipEventRecord(ctx->ev_mutate, st));
        ctx->mutate_pending_steps = n_steps;
        return ASMC_OK;
    }
    return mutate_flow_collect(ctx, n_steps, rho_inout_host, n_accept_host, rho_hist_host, st);
}

extern "C" {

int asmc_pcn_mutate_flow(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq,
                         const asmc_pcn_params* prm, const asmc_coupling* flow, void* work_dev, int64_t work_bytes,
                         int n_steps, uint32_t step0, double* rho_inout_host, int64_t* n_accept_host,
                         double* rho_hist_host, asmc_stream stream) {
    ASMC_REQUIRE(ctx && x && prm && flow, "null pointer");
    ASMC_REQUIRE(flow->affine == ASMC_AFFINE_TANH || (flow->affine == ASMC_AFFINE_SOFTCLIP && flow->kind == ASMC_FLOW_MAF),
                 "bad affine form (the soft-clipped form is for autoregressive flows)");
    // Fewer than 32 dimensions: the one-kernel step on a zero-padded copy (pcn_mutate_padded's scheme: padded rows and tables,
    // the identity beyond d, no noise there) wherever the flow's shape is one the kernel takes - any even d for a coupling flow,
    // any d for an autoregressive one.  Round 3 ran propose / flow / accept kernels at d = 8 / 16 (0.35 / 0.42 ms per step
    // at 1M particles) and the x-state split path at the other d (0.8 ms at d = 20).
    // More than 32 dimensions (round 5): the one-kernel step on 16-particle groups at D = 64 / 128 (asmc_flow16.hip), the
    // problem zero-padded to D when it is narrower.  Round 4 ran propose / flow / targets / accept / copy kernels at 32 < d <= 64
    // with a coupling flow (1.2 - 1.8 ms per step at 1M particles) and had no device path for an autoregressive flow there.
    // (... and at d <= 32 when the flow itself lives in that layout: asmc_flow_layout = 1, an autoregressive flow of hidden width 128)
    const bool lay16 = asmc_flow_layout(flow->kind, flow->dims, flow->hidden) == 1;
    if ((prm->d > 32 || lay16) && prm->d <= 128 && prm->d == flow->dims && ctx->d_mmtab && !getenv("ASMC_PCN_GENERIC") && !getenv("ASMC_PCN_NOPAD") &&
        !getenv("ASMC_PCN_XSTATE")) {
        const int d = prm->d, D = d <= 64 ? 64 : 128;
        asmc_pcn_params p16 = *prm;
        p16.d = D;
        if ((D <= ctx->d_max_pad || d <= 32) && asmc_pcn_flow16_ok(&p16, flow)) {
            ASMC_REQUIRE(ll && lp && lq && rho_inout_host && n_accept_host, "null pointer");
            ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
            ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
            ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference-Gaussian pointer");
            int rc = check_mixture(prm->log_likelihood);
            if (!rc) rc = check_mixture(prm->log_prior);
            if (rc) return rc;
            hipStream_t st = as_stream(stream);
            if (d == D && ((uintptr_t)x % 16) == 0)
                return pcn_mutate_flow16_impl(ctx, n, x, ll, lp, lq, prm, flow, n_steps, step0, rho_inout_host, n_accept_host, rho_hist_host,
                                              stream, 0);
            const size_t es = prm->x_dtype == ASMC_F64 ? 8 : 4;
            const size_t tab_doubles = (size_t)D + 2 * (size_t)D * D + 3 * 2 * (size_t)ASMC_MAX_COMPONENTS * D;
            const size_t tab_bytes = ((tab_doubles * 8 + 255) / 256) * 256;
            const size_t need = tab_bytes + (size_t)n * D * es;
            if (need > ctx->xpad_bytes) {
                ASMC_HIP(hipStreamSynchronize(st));
                if (ctx->d_xpad) (void)hipFree(ctx->d_xpad);
                ctx->d_xpad = nullptr;
                ctx->xpad_bytes = 0;
                if (hipMalloc(&ctx->d_xpad, need) != hipSuccess) {
                    (void)hipGetLastError();
                    asmc_set_error("pcn: no device memory for the zero-padded copy of the state (%zu bytes)", need);
                    return ASMC_ERR_NOMEM;
                }
                ctx->xpad_bytes = need;
            }
            double* tab = reinterpret_cast<double*>(ctx->d_xpad);
            void* xp = reinterpret_cast<char*>(ctx->d_xpad) + tab_bytes;
            PcnDev src;
            memset(&src, 0, sizeof(src));
            src.mu = prm->mu_dev, src.L = prm->L_dev, src.Linv = prm->Linv_dev;
            src.ll = to_dev(prm->log_likelihood), src.lp = to_dev(prm->log_prior);
            src.lq = src.lp;  // (placeholder: the proposal density is the flow)
            ASMC_LAUNCH(ctx, st, "k_pad_tables", k_pad_tables, dim3(1), dim3(256), 0, st, d, D, src, tab);
            ASMC_LAUNCH_CHECK();
            const int grid = grid_for(n * D, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
            if (es == 8)
                ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)x, (double*)xp);
            else
                ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)x, (float*)xp);
            ASMC_LAUNCH_CHECK();
            p16.mu_dev = tab;
            p16.L_dev = tab + D;
            p16.Linv_dev = p16.L_dev + (size_t)D * D;
            asmc_mixture* mix[2] = {&p16.log_likelihood, &p16.log_prior};
            for (int k = 0; k < 2; k++) {
                mix[k]->mu_dev = p16.Linv_dev + (size_t)D * D + (size_t)k * 2 * ASMC_MAX_COMPONENTS * D;
                mix[k]->prec_dev = mix[k]->mu_dev + (size_t)ASMC_MAX_COMPONENTS * D;
            }
            const int caller_defers = ctx->mutate_defer;  // (the un-padding copy goes in front of the read-back's wait: see below)
            ctx->mutate_defer = 1;
            rc = pcn_mutate_flow16_impl(ctx, n, xp, ll, lp, lq, &p16, flow, n_steps, step0, rho_inout_host, n_accept_host, rho_hist_host,
                                        stream, d);
            ctx->mutate_defer = caller_defers;
            if (rc) return rc;
            if (es == 8)
                ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)xp, (double*)x);
            else
                ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)xp, (float*)x);
            ASMC_LAUNCH_CHECK();
            if (caller_defers) return ASMC_OK;
            ctx->mutate_pending_steps = 0;
            return mutate_flow_collect(ctx, n_steps, rho_inout_host, n_accept_host, rho_hist_host, st);
        }
    }
    asmc_pcn_params p2 = *prm;
    p2.d = 32;
    if (prm->d >= 2 && prm->d < 32 && prm->d == flow->dims && 32 <= ctx->d_max_pad && !getenv("ASMC_PCN_GENERIC") &&
        !getenv("ASMC_PCN_NOPAD") && asmc_pcn_flow_fused_ok(&p2, flow) && !getenv("ASMC_PCN_AOS") && !getenv("ASMC_PCN_XSTATE")) {
        ASMC_REQUIRE(ll && lp && lq && rho_inout_host && n_accept_host && work_dev, "null pointer");
        ASMC_REQUIRE(n > 0 && n <= ctx->n_max, "n out of range for this ctx");
        ASMC_REQUIRE(prm->x_dtype == ASMC_F64 || prm->x_dtype == ASMC_F32, "bad x_dtype");
        ASMC_REQUIRE(prm->mu_dev && prm->L_dev && prm->Linv_dev, "null reference-Gaussian pointer");
        ASMC_REQUIRE(work_bytes >= asmc_pcn_flow_work_bytes(n, prm->d, prm->x_dtype), "work buffer too small");  // (as documented, whichever path serves the call)
        int rc = check_mixture(prm->log_likelihood);
        if (!rc) rc = check_mixture(prm->log_prior);
        if (rc) return rc;
        hipStream_t st = as_stream(stream);
        const int d = prm->d, D = 32;
        const size_t es = prm->x_dtype == ASMC_F64 ? 8 : 4;
        const size_t tab_doubles = (size_t)D + 2 * (size_t)D * D + 3 * 2 * (size_t)ASMC_MAX_COMPONENTS * D;
        const size_t tab_bytes = ((tab_doubles * 8 + 255) / 256) * 256;
        const size_t need = tab_bytes + (size_t)n * D * es;
        if (need > ctx->xpad_bytes) {
            ASMC_HIP(hipStreamSynchronize(st));
            if (ctx->d_xpad) (void)hipFree(ctx->d_xpad);
            ctx->d_xpad = nullptr;
            ctx->xpad_bytes = 0;
            if (hipMalloc(&ctx->d_xpad, need) != hipSuccess) {
                (void)hipGetLastError();
                asmc_set_error("pcn: no device memory for the zero-padded copy of the state (%zu bytes)", need);
                return ASMC_ERR_NOMEM;
            }
            ctx->xpad_bytes = need;
        }
        double* tab = reinterpret_cast<double*>(ctx->d_xpad);
        void* xp = reinterpret_cast<char*>(ctx->d_xpad) + tab_bytes;
        PcnDev src;
        memset(&src, 0, sizeof(src));
        src.mu = prm->mu_dev, src.L = prm->L_dev, src.Linv = prm->Linv_dev;
        src.ll = to_dev(prm->log_likelihood), src.lp = to_dev(prm->log_prior);
        src.lq = src.lp;  // (placeholder: the proposal density is the flow)
        ASMC_LAUNCH(ctx, st, "k_pad_tables", k_pad_tables, dim3(1), dim3(256), 0, st, d, D, src, tab);
        ASMC_LAUNCH_CHECK();
        const int grid = grid_for(n * D, ASMC_BLOCK * 4, ASMC_MAX_BLOCKS * 2);
        if (es == 8)
            ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)x, (double*)xp);
        else
            ASMC_LAUNCH(ctx, st, "k_pad_rows", k_pad_rows<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)x, (float*)xp);
        ASMC_LAUNCH_CHECK();
        p2.mu_dev = tab;
        p2.L_dev = tab + D;
        p2.Linv_dev = p2.L_dev + (size_t)D * D;
        asmc_mixture* mix[2] = {&p2.log_likelihood, &p2.log_prior};
        for (int k = 0; k < 2; k++) {
            mix[k]->mu_dev = p2.Linv_dev + (size_t)D * D + (size_t)k * 2 * ASMC_MAX_COMPONENTS * D;
            mix[k]->prec_dev = mix[k]->mu_dev + (size_t)ASMC_MAX_COMPONENTS * D;
        }
        // The unpadding copy must sit behind the mutation on the stream AND in front of the read-back's synchronisation: the
        // blocking form returns with x final (it used to synchronise inside the impl and enqueue the copy after
        // that, so a C caller reading x on another stream saw the padded run's input).  So the impl always runs deferred
        // here, the copy follows, and the blocking form collects afterwards.
        const int caller_defers = ctx->mutate_defer;
        ctx->mutate_defer = 1;
        rc = pcn_mutate_flow_impl(ctx, n, xp, ll, lp, lq, &p2, flow, work_dev, work_bytes, n_steps, step0, rho_inout_host, n_accept_host,
                                  rho_hist_host, stream, d);
        ctx->mutate_defer = caller_defers;
        if (rc == ASMC_ERR_UNSUPPORTED) {
            // the one-kernel step declined after the pre-check (no register path for this pointer, no coordinate-major state): nothing
            // of the caller's has been touched yet (only the padded copy was whitened) - the unpadded path serves the shape, as before
            ctx->mutate_pending_steps = 0;
            return pcn_mutate_flow_impl(ctx, n, x, ll, lp, lq, prm, flow, work_dev, work_bytes, n_steps, step0, rho_inout_host,
                                        n_accept_host, rho_hist_host, stream, 0);
        }
        if (rc) return rc;
        if (es == 8)
            ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<double>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const double*)xp, (double*)x);
        else
            ASMC_LAUNCH(ctx, st, "k_unpad_rows", k_unpad_rows<float>, dim3(grid), dim3(ASMC_BLOCK), 0, st, n, d, D, (const float*)xp, (float*)x);
        ASMC_LAUNCH_CHECK();
        if (caller_defers) return ASMC_OK;  // asmc_pcn_mutate_flow_result waits for the mutation's event; the copy is stream-ordered behind it
        ctx->mutate_pending_steps = 0;
        // (stream synchronisation, not the event recorded in front of the copy: x must be final when this call returns)
        return mutate_flow_collect(ctx, n_steps, rho_inout_host, n_accept_host, rho_hist_host, st);
    }
    return pcn_mutate_flow_impl(ctx, n, x, ll, lp, lq, prm, flow, work_dev, work_bytes, n_steps, step0, rho_inout_host, n_accept_host,
                                rho_hist_host, stream, 0);
}

int asmc_pcn_mutate_flow_enqueue(asmc_ctx* ctx, int64_t n, void* x, double* ll, double* lp, double* lq,
                                 const asmc_pcn_params* prm, const asmc_coupling* flow, void* work_dev, int64_t work_bytes,
                                 int n_steps, uint32_t step0, double rho, asmc_stream stream) {
    ASMC_REQUIRE(ctx != nullptr, "null ctx");
    ASMC_REQUIRE(ctx->mutate_pending_steps == 0, "a deferred mutation is still pending (asmc_pcn_mutate_flow_result)");
    int64_t dummy_acc = 0;
    ctx->mutate_defer = 1;
    const int rc = asmc_pcn_mutate_flow(ctx, n, x, ll, lp, lq, prm, flow, work_dev, work_bytes, n_steps, step0, &rho, &dummy_acc,
                                        nullptr, stream);
    ctx->mutate_defer = 0;
    return rc;
}

int asmc_pcn_mutate_flow_result(asmc_ctx* ctx, int n_steps, double* rho_out_host, int64_t* n_accept_host, double* rho_hist_host,
                                asmc_stream stream) {
    ASMC_REQUIRE(ctx && rho_out_host && n_accept_host, "null pointer");
    ASMC_REQUIRE(ctx->mutate_pending_steps == n_steps && n_steps > 0, "no deferred mutation of this length is pending");
    ctx->mutate_pending_steps = 0;
    return mutate_flow_collect(ctx, n_steps, rho_out_host, n_accept_host, rho_hist_host, as_stream(stream), ctx->ev_mutate);
}

int64_t asmc_pcn_flow_nonfinite(asmc_ctx* ctx) { return ctx ? (int64_t)ctx->flow_nonfinite : -1; }

int64_t asmc_pcn_lq_nan(asmc_ctx* ctx) { return ctx ? (int64_t)ctx->lq_nan : -1; }

}  // extern "C"
