// asmc_pcn_dev.h — device helpers shared by the pCN kernels (asmc_pcn.hip: d <= 32 register-resident and generic
// kernels; asmc_pcn_mm.hip: d = 64 / 128 on the fp64 matrix cores): counter-based noise, tempered target, parameter
// structs.
#pragma once
#include "asmc_common.h"

bool asmc_flow_math_split();  // asmc_flow.hip

// step-size adaptation closed by the last block of a step's kernel (k_pcn_adapt's arithmetic); done == NULL: left to a
// k_pcn_adapt launch.  Sharded runs exchange the accept counts between ranks first: the last block leaves the rank's
// count in `cell`, the exchange hook sums it over the ranks on the stream, and the NEXT step's kernel adapts the step
// size in its prologue (`prev_cell`: every block derives the same rho from rho_hist[t-1] and the global count, block 0
// records them), so a step boundary is the exchange alone; a k_pcn_adapt launch closes the last step of a call.
struct PcnAdaptArgs {
    unsigned int* done;     // zeroed arrival counter of this step
    unsigned long long* nonfinite;  // += proposals whose flow density came out non-finite (rejected); may be NULL
    long long* cell;        // sharded runs: the rank's count goes HERE; NULL: this kernel closes the step itself
    const long long* prev_cell;  // sharded runs, t > 0: the GLOBAL count of step t-1 (every block reads it BEFORE it
                                 // arrives at `done`, the last block overwrites `cell` after all have arrived)
    long long* counts_out;  // [t] <- accepted particles of the step
    double* rho;            // step size: read by this step, adapted for the next
    double* rho_hist;       // [t] <- the step size this step used
    double target;
    int64_t n;              // population the acceptance rate refers to (sharded: the global one)
    int t, adapt;           // adapt: 0 off, 1 after every step, k >= 2 lagged by blocks of k steps (k_pcn_adapt, asmc_pcn.hip)
    int last;               // t is the final step of the call (a lagged adaptation closes its open block there)
};

// log rho += (acc - target)/(t+1)^0.75, rho clipped to [1e-4, 0.99]; one definition for every kernel that
// adapts, so that the ranks of a sharded run and the single-rank kernels agree to the bit
__device__ __forceinline__ double pcn_adapt_rho(double rho, long long c, int64_t n, double target, int t) {
    const double acc = (double)c / (double)n;
    double r = exp(log(rho) + (acc - target) / pow((double)(t + 1), 0.75));
    r = r < 1e-4 ? 1e-4 : r;
    r = r > 0.99 ? 0.99 : r;
    return r;
}

// =============================================================================================
// Philox4x32-10 (Salmon et al. SC'11; Random123 constants) and Box-Muller
// =============================================================================================
#ifndef ASMC_PHILOX_ROUNDS
#define ASMC_PHILOX_ROUNDS 10  // (the specification; other values exist for timing experiments only: tools/build_variant.sh)
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < ASMC_PHILOX_ROUNDS; r++) {
#ifdef ASMC_PHILOX_MULHI  // (round 1-3 form: two multiply instructions per product)
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#else  // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32): half the integer multiplies of a block
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#endif
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ __forceinline__ double u01_from_words(uint32_t hi, uint32_t lo) {
    unsigned long long v = ((unsigned long long)hi << 21) ^ ((unsigned long long)lo >> 11);
    v &= ((1ULL << 53) - 1ULL);
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// ---- fp64 elementary functions of the Box-Muller transform on their restricted domains -----------------------------
// The noise is most of the vector-ALU work of every pCN kernel; the general-purpose library versions
// carry range checks, special cases and argument reductions these arguments never need.  Algorithms and coefficients:
// FreeBSD msun e_log.c / k_sin.c / k_cos.c (each < 1 ulp); -ffp-contract=off, so every fma below is written out.

// log(u) for a positive normal u (the Box-Muller radius feeds it [2^-54, 1], the logit transform (1, 2])
__device__ __forceinline__ double bm_log_unit(double u) {
    double m = __builtin_amdgcn_frexp_mant(u);  // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(u);
    const bool lowhalf = m < 0.70710678118654752440;
    m = lowhalf ? m + m : m;  // sqrt(1/2) <= m < sqrt(2)
    k = lowhalf ? k - 1 : k;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    // s = f / d: v_rcp_f64 refined twice is far below the 1e-18 this term needs
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double s = f * r;
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// sqrt(a), 1e-17 < a < 1e3: v_rsq_f64 + one coupled Goldschmidt step + the final residual correction
__device__ __forceinline__ double bm_sqrt(double a) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double dres = fma(-g, g, a);
    return a > 0.0 ? fma(dres, h, g) : 0.0;
}

// (sin, cos)(2 pi u), u in (0, 1]: quarter-turn reduction (exact), kernels on |x| <= pi/4
__device__ __forceinline__ void bm_sincos_turns(double u, double& sn, double& cs) {
    const double q = __builtin_rint(4.0 * u);  // 0 .. 4
    const double f = fma(q, -0.25, u);         // exact, |f| <= 1/8
    const double x = f * 6.28318530717958647692;
    const double z = x * x;
    const double v = z * x;
    const double rs = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                 -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double sx = fma(v, fma(z, rs, -1.66666666666666324348e-01), x);
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                 -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double wv = 1.0 - hz;
    const double cx = wv + fma(z, rc, (1.0 - wv) - hz);
    const int qi = (int)q;
    // angle = x + q pi/2: q = 0 (s, c); 1 (c, -s); 2 (-s, -c); 3 (-c, s); 4 = 0
    const bool swap = qi & 1;
    const double s0 = swap ? cx : sx, c0 = swap ? sx : cx;
    sn = (qi & 2) ? -s0 : s0;
    cs = ((qi + 1) & 2) ? -c0 : c0;
}

// two uniforms in (0, 1] -> two standard normals
__device__ __forceinline__ void box_muller(double u1, double u2, double& z0, double& z1) {
#ifdef ASMC_NOISE_LIBM  // the device library's log / sqrt / sincospi instead (diagnostic: same values to an ulp or two)
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
#else
    const double r = bm_sqrt(-2.0 * bm_log_unit(u1));
    double s, c;
    bm_sincos_turns(u2, s, c);
#endif
    z0 = r * c;
    z1 = r * s;
}

// ---- the default ("f64") noise: table-driven fp64 Box-Muller on 32-bit uniforms (round 3) ---------------------------
// One Philox block = FOUR standard normals (two Box-Muller pairs).  Words (w0, w1) make pair A, (w2, w3) pair B:
//     u = (w_r + 1/2) 2^-32  in (0, 1),   t = (w_a + 1/2) 2^-32 turns,   r = sqrt(-2 ln u),   (z_even, z_odd) = r (cos, sin)(2 pi t)
// all in fp64.  32-bit uniforms: |z| <= 6.76, P(|z| > 6.66) = 2.7e-11 is the mass a 53-bit radius would add; the angle has
// 2^32 directions.  (Rounds 1-2 spent one block per PAIR on two 53-bit uniforms and evaluated log / sincos with msun's
// general kernels: 179 vector instructions per pair, 45 % of the fused flow step's vector work.)  The two elementary
// functions are table driven (bm_tab: 6 KB, staged into LDS by every kernel that draws):
//   * ln: X = w_r + 1/2 = 2^e m, m in [1/2, 1); interval i = top 7 mantissa bits of m, c_i its centre, rc_i = fl(1 / c_i);
//     r = fma(m, rc_i, -1) (|r| <= 2^-8, one rounding) and -2 ln u = (32 - e) 2 ln 2 + 2 ln(rc_i) - 2 log1p(r) with the
//     degree-6 series of log1p (truncation 2^-56 / 3.5); the table holds (rc_i, 2 ln rc_i).
//   * sincos: the top 8 bits of w_a pick (S, C) = (sin, cos)(2 pi (k + 1/2) / 256) from the table, the other 24 bits are
//     x = 2 pi ((w_a mod 2^24) + 1/2 - 2^23) 2^-32, |x| <= 2 pi / 512, whose sine and cosine - 1 are 4- and 3-term series
//     (truncation 1.7e-23, 1.3e-20); angle addition S + (S (cos x - 1) + C sin x), C + (C (cos x - 1) - S sin x).
// Against libm on the same (u, t) (the CPU restatement used by the tests): a few 1e-16 absolute.
// (BM_SC_N, BM_LG_N, BM_TAB_N: asmc_common.h)
typedef double bm_d2 __attribute__((ext_vector_type(2)));
void asmc_bm_table_host(double* tab);  // asmc_ctx.hip

template <int THREADS>
__device__ __forceinline__ void bm_tab_stage(bm_d2* __restrict__ s, const double* __restrict__ g) {
    for (int e = threadIdx.x; e < BM_TAB_N; e += THREADS) s[e] = reinterpret_cast<const bm_d2*>(g)[e];
}
__device__ __forceinline__ void bm_tab_stage_rt(bm_d2* __restrict__ s, const double* __restrict__ g) {
    for (int e = threadIdx.x; e < BM_TAB_N; e += (int)blockDim.x) s[e] = reinterpret_cast<const bm_d2*>(g)[e];
}
// the calling kernel's own 6 KB of LDS for the tables (one allocation per kernel that calls it; the caller stages
// and synchronises)
__device__ __forceinline__ bm_d2* bm_lds() {
    __shared__ bm_d2 s_bm[BM_TAB_N];
    return s_bm;
}

// one Box-Muller pair from a radius word and an angle word, in three stages so that callers with several pairs in flight
// can issue every table read before the first use (LDS latency): indices -> table-free series -> combination
struct BmPair {
    double m, sx, cm1;  // mantissa of w_r + 1/2; sin x and cos x - 1 of the angle residual
    int e;              // its exponent
};
__device__ __forceinline__ void bm_pair_idx(uint32_t wr, uint32_t wa, BmPair& t, uint32_t& i_sc, uint32_t& i_lg) {
    i_sc = wa >> 24;
    const double X = (double)wr + 0.5;
    t.m = __builtin_amdgcn_frexp_mant(X);  // [1/2, 1)
    t.e = __builtin_amdgcn_frexp_exp(X);   // 0 .. 32
    const uint32_t mh = (uint32_t)((unsigned long long)__double_as_longlong(t.m) >> 32);
    i_lg = BM_SC_N + ((mh >> 13) & (BM_LG_N - 1));
    const double x = fma((double)((int)(wa & 0xFFFFFFu) - (1 << 23)), 1.4629180792671596e-09, 7.314590396335798e-10);  // 2 pi 2^-32, pi 2^-32
    const double z = x * x;
    const double ps = fma(z, fma(z, -1.9841269841269841e-04, 8.3333333333333332e-03), -1.6666666666666666e-01);
    t.sx = fma(x * z, ps, x);
    t.cm1 = z * fma(z, fma(z, -1.3888888888888889e-03, 4.1666666666666664e-02), -0.5);
}
__device__ __forceinline__ void bm_pair_fin(const BmPair& t, const bm_d2 sc, const bm_d2 lg, double& z0, double& z1) {
    const double sn = sc.x + fma(sc.y, t.sx, sc.x * t.cm1);
    const double cs = sc.y + fma(-sc.x, t.sx, sc.y * t.cm1);
    const double r = fma(t.m, lg.x, -1.0);
    double p = fma(r, 3.3333333333333331e-01, -4.0000000000000002e-01);
    p = fma(r, p, 0.5);
    p = fma(r, p, -6.6666666666666663e-01);
    p = fma(r, p, 1.0);
    p = fma(r, p, -2.0);
    const double a = fma((double)(32 - t.e), 1.3862943611198906, lg.y) + r * p;  // -2 ln u >= 2.3e-10
    // sqrt(a): v_rsq_f64 + one coupled Goldschmidt step + the residual correction (bm_sqrt without its a <= 0 guard)
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g);
    h = fma(h, rr, h);
    const double rad = fma(fma(-g, g, a), h, g);
    z0 = rad * cs;
    z1 = rad * sn;
}
__device__ __forceinline__ void bm_pair32(uint32_t wr, uint32_t wa, const bm_d2* __restrict__ tab, double& z0, double& z1) {
    BmPair t;
    uint32_t i_sc, i_lg;
    bm_pair_idx(wr, wa, t, i_sc, i_lg);
    bm_pair_fin(t, tab[i_sc], tab[i_lg], z0, z1);
}

// counter = {gid_lo, gid_hi, step, slot}; key = {seed_lo, seed_hi}.  Coordinates 4 q .. 4 q + 3 come from the block with
// slot = q | 0x20000000.
__device__ __forceinline__ void normal_quad(unsigned long long seed, unsigned long long gid, uint32_t step, uint32_t q,
                                            const bm_d2* __restrict__ tab, double& z0, double& z1, double& z2, double& z3) {
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), step, q | 0x20000000u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    BmPair ta, tb;
    uint32_t ia_sc, ia_lg, ib_sc, ib_lg;
    bm_pair_idx(w[0], w[1], ta, ia_sc, ia_lg);
    bm_pair_idx(w[2], w[3], tb, ib_sc, ib_lg);
    const bm_d2 sca = tab[ia_sc], lga = tab[ia_lg], scb = tab[ib_sc], lgb = tab[ib_lg];
    bm_pair_fin(ta, sca, lga, z0, z1);
    bm_pair_fin(tb, scb, lgb, z2, z3);
}
// coordinates 2 pr, 2 pr + 1 alone (a whole block for one pair: callers that own single pairs)
__device__ __forceinline__ void normal_pair(unsigned long long seed, unsigned long long gid, uint32_t step, uint32_t pr,
                                            const bm_d2* __restrict__ tab, double& z0, double& z1) {
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), step, (pr >> 1) | 0x20000000u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const bool second = pr & 1u;
    bm_pair32(second ? w[2] : w[0], second ? w[3] : w[1], tab, z0, z1);
}

// Fast noise (NOISE_F32): one Philox block -> FOUR standard normals through fp32 Box-Muller on the
// hardware transcendental units (v_log_f32 / v_sin_f32 / v_cos_f32 / v_sqrt_f32):
//   u = w*2^-32 + 2^-33 (float), t = w'*2^-32 revolutions, r = sqrt(-2 ln u), z = r (cos 2 pi t, sin 2 pi t).
// ~1e-7 relative accuracy, tails to 6.7 sigma; 8x fewer VALU cycles than the fp64 path.
// the four fp32 normals of Philox block `slot` (fast-noise mode); normal_quad_f32 widens them
__device__ __forceinline__ void normal_quad_f32_raw(unsigned long long seed, unsigned long long gid, uint32_t step,
                                                    uint32_t slot, float& f0, float& f1, float& f2, float& f3) {
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), step, slot | 0x40000000u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), w);
    const float k = 2.3283064365386963e-10f;  // 2^-32
    const float u0 = fmaf((float)w[0], k, 1.1641532182693481e-10f);
    const float u1 = fmaf((float)w[2], k, 1.1641532182693481e-10f);
    const float t0 = (float)w[1] * k, t1 = (float)w[3] * k;
    // ln u = log2(u) * ln 2
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    f0 = r0 * __builtin_amdgcn_cosf(t0);
    f1 = r0 * __builtin_amdgcn_sinf(t0);
    f2 = r1 * __builtin_amdgcn_cosf(t1);
    f3 = r1 * __builtin_amdgcn_sinf(t1);
}

__device__ __forceinline__ void normal_quad_f32(unsigned long long seed, unsigned long long gid, uint32_t step,
                                                uint32_t slot, double& z0, double& z1, double& z2, double& z3) {
    float f0, f1, f2, f3;
    normal_quad_f32_raw(seed, gid, step, slot, f0, f1, f2, f3);
    z0 = (double)f0;
    z1 = (double)f1;
    z2 = (double)f2;
    z3 = (double)f3;
}

__device__ __forceinline__ double accept_uniform(unsigned long long seed, unsigned long long gid,
                                                 uint32_t step) {
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), step, 0xFFFFFFFFu, (uint32_t)seed,
                  (uint32_t)(seed >> 32), w);
    return u01_from_words(w[0], w[1]);
}

// ---- t-preconditioned Crank-Nicolson (step_fn "tpcn"; specification: DESIGN.md §3.6) ---------------------------
// unit-scale Gamma(shape >= 1) variate, Marsaglia & Tsang (2000): ONE Philox block per attempt a < 8 - the normal from words
// 0, 1 (the first variate of bm_pair32 in the default mode, slot 0x80000000 | a; the hardware fp32 Box-Muller of
// normal_quad_f32 in the fast-noise mode F32, slot 0xC0000000 | a), the 53-bit uniform from words 2, 3.
template <bool F32 = false>
__device__ __forceinline__ double gamma_unit(double shape, unsigned long long seed, unsigned long long gid,
                                             uint32_t step, const bm_d2* __restrict__ tab) {
    const double dd = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * dd);
    // fully unrolled (nested early exits, no loop).  P(all 8 attempts fail) < 0.05^8 = 4e-11.
#pragma unroll
    for (uint32_t a = 0; a < 8; a++) {
        double z0;
        uint32_t w[4];
        philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), step, (F32 ? 0xC0000000u : 0x80000000u) | a, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        if (F32) {
            const float k = 2.3283064365386963e-10f;  // 2^-32
            const float u0 = fmaf((float)w[0], k, 1.1641532182693481e-10f);
            const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));
            z0 = (double)(r0 * __builtin_amdgcn_cosf((float)w[1] * k));
        } else {
            double z1;
            bm_pair32(w[0], w[1], tab, z0, z1);
        }
        const double u = u01_from_words(w[2], w[3]);
        double v = 1.0 + c * z0;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(u) < 0.5 * z0 * z0 + dd - dd * v + dd * log(v)) return dd * v;
    }
    return dd;
}
// proposal scale rho sqrt(s), s = 1/G, G = g * 2/(nu + q0) with the unit-scale variate g = gam[i] drawn by k_gamma_draw
// just before the step kernel.  (The rejection sampler lives in its own small kernel: any retry control flow inside the
// register-resident / matrix-core step kernels makes LLVM hoist and spill their ~1300 invariant scalar loads; those
// kernels also select the Student-t form at compile time, TP.)  gam == nullptr: Gaussian reference (rho).
template <bool TP = true>
__device__ __forceinline__ double tpcn_scale(double rho, double nu, double q0, const double* __restrict__ gam, int64_t i) {
    if (!TP || gam == nullptr) return rho;
    return rho * sqrt((nu + q0) / (2.0 * gam[i]));
}
// minus the log-density of the reference at |y|^2 = q, up to a constant: q/2 (Gaussian) or ((d + nu)/2) log(1 + q/nu)
// compile-time forms for the register-resident / matrix-core kernels: straight-line code, no test of gam or nu
template <bool TP>
__device__ __forceinline__ double tpcn_scale_ct(double rho, double nu, double q0, const double* __restrict__ gam, int64_t i) {
    if (!TP) return rho;
    return rho * sqrt((nu + q0) / (2.0 * gam[i]));
}
// log(1 + x) for x >= 0 - the Student-t reference's correction, x = |y|^2 / nu, twice per particle and step of the reference's default
// step (tpCN) - by the noise generator's table-free log on fl(1 + x) (round 6).  The rounding of 1 + x is an absolute 1.1e-16 in the
// result, which enters the accept test through a difference of O(1) terms; the device library's log1p spends about 150 vector
// instructions on ranges and signs this argument never has.  Non-finite arguments propagate as log1p propagates them.
__device__ __forceinline__ double log1p_nonneg(double x) {
    const double u = 1.0 + x;
    const double r = bm_log_unit(u);
    return u < INFINITY ? r : u;
}
template <bool TP>
__device__ __forceinline__ double ref_corr_ct(double q, double nu, int d) {
    if (!TP) return 0.5 * q;
    return 0.5 * ((double)d + nu) * log1p_nonneg(q / nu);
}

template <bool TP = true>
__device__ __forceinline__ double ref_corr(double q, double nu, int d) {
    if (!TP) return 0.5 * q;
    return nu > 0.0 ? 0.5 * ((double)d + nu) * log1p_nonneg(q / nu) : 0.5 * q;
}

struct MixDev {
    int C;
    const double* logw;
    const double* mu;
    const double* prec;
};

static inline MixDev to_dev(const asmc_mixture& m) {
    MixDev r;
    r.C = m.n_components;
    r.logw = m.logw_dev;
    r.mu = m.mu_dev;
    r.prec = m.prec_dev;
    return r;
}


// samples.py:1217-1219 + smc/base.py:507-519: the tempered log-target, NaN -> -inf as the reference maps it.  +inf -> -inf
// as well (round 6): the reference's integration test with a +inf likelihood hole (tests/integration_tests/
// test_integration.py:131-166) finishes, so the third-party step it calls cannot let a particle settle on a +inf
// log-target (it would carry ll = +inf, the next log-sum-exp is NaN and the temperature search stalls).  This repository's
// reading of that unverified behaviour: a proposal whose tempered log-target is +inf or NaN is rejected (and counted as a
// rejection); a carried +inf (only a caller-supplied state can hold one) loses against any finite proposal.
__device__ __forceinline__ double log_p_t_guard(double r) { return (r < INFINITY) ? r : -INFINITY; }
__device__ __forceinline__ double log_p_t(double ll, double lp, double lq, double beta) {
    return log_p_t_guard((1.0 - beta) * lq + beta * (ll + lp));
}


struct PcnDev {
    int d;
    double beta;
    const double* mu;
    const double* L;
    const double* Linv;
    MixDev ll, lp, lq;
    unsigned long long seed, gid0;
    int noise;
    double nu;  // > 0: Student-t reference with nu degrees of freedom (tpCN); <= 0: Gaussian reference (pCN)
    const double* gam;  // [n] unit-scale Gamma((d + nu)/2) variates of the current step (tpCN), else nullptr
    void* ys;           // coordinate-major whitened state of the register-resident kernels (or nullptr)
    long long n_pad;
    const double* bmtab;  // the Box-Muller tables in HBM (ctx->d_bmtab)
    unsigned char* tile_par;  // fused flow step: which half of the state allocation holds each 64-particle tile (else nullptr)
    int d_noise;        // > 0: a d_noise-dimensional problem zero-padded to d (asmc_pcn_mutate): coordinates >= d_noise get no noise
    int dpad;           // > d: run the d-dimensional problem on the kernels compiled for dpad (identity-padded tables)
    int mode;  // PCN_X_STEP / PCN_Y_STEP / PCN_WHITEN / PCN_UNWHITEN (register-resident kernels)
};


// ---- packed parameter block and coordinate-major state access of the register-resident kernels (d <= 32) ---------
// layout (doubles): Ltri[D(D+1)/2] | Linvtri[D(D+1)/2] | mu[D] | 3 x { logw[8] | mu[8*D] | prec[8*D] }  (ll, lp, lq)
#define PTAB_TRI(D) ((D) * ((D) + 1) / 2)
#define PTAB_MIX(D) (ASMC_MAX_COMPONENTS * (1 + 2 * (D)))
#define PTAB_SIZE(D) (2 * PTAB_TRI(D) + (D) + 3 * PTAB_MIX(D))
// behind it (round 6): L once more, in the order the coordinate-major step consumes it - the 4 x 4 blocks of the lower block triangle,
// row group by row group, block (g, c) at 16 (g (g + 1) / 2 + c), entry [k % 4][r] = L[4 g + r][4 c + k % 4] (zero above the diagonal)
#define PTAB_BLK(D) (((D) / 4) * ((D) / 4 + 1) / 2 * 16)
struct PcnScalars {
    double beta;
    double nu;  // Student-t degrees of freedom of the reference (tpCN) or <= 0 (Gaussian pCN)
    const double* gam;  // per-particle Gamma((d + nu)/2, 1) variates of this step (k_gamma_draw), nullptr for pCN
    void* ys;           // coordinate-major whitened state (PCN_*_S modes)
    long long n_pad;    // its row length (n rounded up to 64)
    int d_real;         // PCN_X_PROPOSE_PAD*: the problem's dimension (< D)
    unsigned long long seed, gid0;
    int c_ll, c_lp, c_lq;
    const double* bmtab;  // the Box-Muller tables in HBM (ctx->d_bmtab)
    unsigned char* tile_par;  // see PcnDev
    int d_noise;              // see PcnDev (0: no padding)
};


// coordinate-major state through buffer instructions: one 128-bit descriptor in SGPRs, the lane as a 32-bit VGPR
// offset and the coordinate's row (j * n_pad + tile) as the scalar offset - instead of one 64-bit VGPR address pair per
// coordinate (32 pairs = 64 VGPRs held from the loads to the conditional stores with plain global accesses)
template <typename T>
__device__ __forceinline__ double soa_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 8) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
        return __builtin_bit_cast(double, v);
    } else {
        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
        return (double)__uint_as_float(v);
    }
}
template <typename T>
__device__ __forceinline__ void soa_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x) {
    if constexpr (sizeof(T) == 8) {
        using u2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), r, (int)voff, (int)soff, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)x), r, (int)voff, (int)soff, 0);
    }
}


// fused flow-proposal step (asmc_pcn_fused.hip)
bool asmc_pcn_flow_fused_ok(const asmc_pcn_params* prm, const asmc_coupling* f);
int asmc_pcn_flow_fused_launch(asmc_ctx* ctx, int64_t n, int x_dtype, double* ll, double* lp, double* lq, const PcnDev& pd,
                               const asmc_coupling* f, const double* rho_ptr, uint32_t step, unsigned int* tile_counter,
                               long long* block_counts, int* grid_out, const PcnAdaptArgs& adapt, hipStream_t st);

// d = 64 / 128 pCN on the fp64 matrix cores (asmc_pcn_mm.hip)
#define MM_WHITEN 0
#define MM_STEP 1
#define MM_UNWHITEN 2
#define MM_STEP_T 3  // MM_STEP with the Student-t reference (tpCN)
#define MM_XPROPOSE 4    // proposal half of the split path on the x-state: y = Linv (x - mu), y' = a y + rho xi, x' = mu + L y' -> x_prop, quadratic forms
#define MM_XPROPOSE_T 5  // ... with the Student-t reference
#define MM_UNWHITEN_X 6  // x = mu + L y only: the carried ll / lp / lq stay (a flow-proposal mutation: log q is not a built-in density)
bool asmc_pcn_mm_supported(int d, const void* x);
int asmc_pcn_mm_pack(asmc_ctx* ctx, const PcnDev& pd, hipStream_t st);
// asmc_flow16.hip: the one-kernel flow-proposal step at d = 64 / 128 (padded), tables built once per mutation
int asmc_pcn_flow16_tables(asmc_ctx* ctx, const PcnDev& pd, const asmc_coupling* f, hipStream_t st);
int asmc_pcn_flow16_launch(asmc_ctx* ctx, int64_t n, int x_dtype, void* x, double* ll, double* lp, double* lq, const PcnDev& pd,
                           const asmc_coupling* f, const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                           unsigned long long* nonfinite, hipStream_t st);
int asmc_pcn_mm_launch(asmc_ctx* ctx, int64_t n, int x_dtype, void* x, double* ll, double* lp, double* lq, const PcnDev& pd,
                       int mode, const double* rho_ptr, uint32_t step, long long* block_counts, int* grid_out,
                       hipStream_t st);
bool asmc_gram_mm_supported(int d, const void* x);
int asmc_gram_mm_launch(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x, const double* d_center, int* grid_out,
                        hipStream_t st, double* out2, double n_div);
