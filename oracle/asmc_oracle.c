/*
 * asmc_oracle.c — CPU restatement of aspire's SMC particle-batch arithmetic.
 *
 * TEST INFRASTRUCTURE.  This file is the *checker*, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load the library built from it.
 * The product path (aspire_amd/) never links, imports or calls it.
 *
 * Parity status
 *   - orc_unnormalized_log_weights .. orc_resample_indices, orc_determine_beta: PINNED against the
 *     real reference (imported in the build container through oracle/ref_shim.py) by the golden
 *     vectors in tests/golden/ref_*.npz (generator: oracle/make_golden.py) and against numpy's
 *     Generator.choice / PCG64 live (numpy is present on the GPU box).
 *   - orc_pcn_*: PARITY UNPINNED.  The reference delegates the mutation kernel to the third-party
 *     package `minipcn[array-api]>=0.2.0a3` (reference pyproject.toml:44; call site
 *     src/aspire/samplers/smc/minipcn.py:89-114), which is absent from /root/reference and from
 *     this image.  These functions restate THIS REPOSITORY's pCN specification.
 *   - orc_coupling_logprob: PARITY UNPINNED.  The reference's flows are zuko modules (reference
 *     pyproject.toml:40, call sites src/aspire/flows/torch/flows.py:156-168,327-387); zuko is absent.
 *     The function restates this repository's CouplingFlow (aspire_amd/flows.py) in fp32 and is
 *     pinned only against that torch module (tests/test_oracle.py).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Plain C11, no dependencies beyond libm.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_NAN (-2)
#define ORC_ERR_ARG (-1)
#define ORC_ERR_PSUM (-3)
#define ORC_ERR_BETA_STALL (-4)

/* ------------------------------------------------------------------------------------------
 * numpy reductions.  np.sum / np.mean / np.var over a contiguous float64 vector use numpy's
 * pairwise summation (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum): blocks of
 * <=128 elements are summed with 8 interleaved accumulators, larger inputs are split in halves
 * (half rounded down to a multiple of 8).  Restated here so that log-sum-exp values agree with the
 * reference to the last bits (they feed `eff >= target` comparisons in the beta bisection).
 * ---------------------------------------------------------------------------------------- */
static double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0; /* numpy starts from -0.0 for the n<8 branch; irrelevant for our inputs */
        res = -0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8) {
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* utils.py:248-255  logsumexp(x): c = x.max(); c + log(sum(exp(x - c))).
 * `tmp` is caller scratch of n doubles.  numpy's max propagates NaN. */
static double logsumexp_tmp(const double* x, int64_t n, double* tmp) {
    double c = x[0];
    int has_nan = isnan(c);
    for (int64_t i = 1; i < n; i++) {
        if (isnan(x[i])) has_nan = 1;
        if (x[i] > c) c = x[i];
    }
    if (has_nan) c = NAN;
    for (int64_t i = 0; i < n; i++) tmp[i] = exp(x[i] - c);
    return c + log(pairwise_sum(tmp, n));
}

double orc_logsumexp(const double* x, int64_t n) {
    if (n <= 0) return NAN;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)n);
    double r = logsumexp_tmp(x, n, tmp);
    free(tmp);
    return r;
}

/* samples.py:1221-1224  SMCSamples.unnormalized_log_weights(beta):
 *   (self.beta - beta) * log_q + (beta - self.beta) * (log_likelihood + log_prior)
 * in exactly that association. */
void orc_unnormalized_log_weights(int64_t n, const double* ll, const double* lp, const double* lq,
                                  double beta0, double beta, double* out) {
    const double c1 = beta0 - beta;
    const double c2 = beta - beta0;
    for (int64_t i = 0; i < n; i++) {
        double t1 = c1 * lq[i];
        double t2 = c2 * (ll[i] + lp[i]);
        out[i] = t1 + t2;
    }
}

/* samples.py:1244-1249  SMCSamples.log_weights(beta): NaN guard -> ValueError; returns
 *   log_w + (logsumexp(log_w) - log(N))      (note: ADDS the log-ratio, a constant shift).
 * returns ORC_ERR_NAN when the reference would raise. */
int orc_log_weights(int64_t n, const double* ll, const double* lp, const double* lq, double beta0,
                    double beta, double* out) {
    if (n <= 0) return ORC_ERR_ARG;
    orc_unnormalized_log_weights(n, ll, lp, lq, beta0, beta, out);
    for (int64_t i = 0; i < n; i++)
        if (isnan(out[i])) return ORC_ERR_NAN;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)n);
    double ratio = logsumexp_tmp(out, n, tmp) - log((double)n);
    free(tmp);
    for (int64_t i = 0; i < n; i++) out[i] = out[i] + ratio;
    return ORC_OK;
}

/* utils.py:510-512  effective_sample_size(log_w) = exp(logsumexp(log_w)*2 - logsumexp(log_w*2)) */
double orc_effective_sample_size(const double* lw, int64_t n) {
    double* tmp = (double*)malloc(sizeof(double) * (size_t)n * 2);
    double* lw2 = tmp + n;
    double l1 = logsumexp_tmp(lw, n, tmp);
    for (int64_t i = 0; i < n; i++) lw2[i] = lw[i] * 2.0;
    double l2 = logsumexp_tmp(lw2, n, tmp);
    free(tmp);
    return exp(l1 * 2.0 - l2);
}

/* ESS of the tempered weights at `beta` (smc/base.py:179-181, :419, :430):
 * effective_sample_size(samples.log_weights(beta)).  *status = ORC_ERR_NAN on the NaN guard. */
double orc_ess_at_beta(int64_t n, const double* ll, const double* lp, const double* lq,
                       double beta0, double beta, int* status) {
    double* lw = (double*)malloc(sizeof(double) * (size_t)n);
    int st = orc_log_weights(n, ll, lp, lq, beta0, beta, lw);
    if (status) *status = st;
    double ess = NAN;
    if (st == ORC_OK) ess = orc_effective_sample_size(lw, n);
    free(lw);
    return ess;
}

/* samples.py:1226-1228  log_evidence_ratio(beta) = logsumexp(unnormalized) - log(N) */
double orc_log_evidence_ratio(int64_t n, const double* ll, const double* lp, const double* lq,
                              double beta0, double beta) {
    double* lw = (double*)malloc(sizeof(double) * (size_t)n * 2);
    orc_unnormalized_log_weights(n, ll, lp, lq, beta0, beta, lw);
    double r = logsumexp_tmp(lw, n, lw + n) - log((double)n);
    free(lw);
    return r;
}

/* samples.py:1230-1242  log_evidence_ratio_variance(beta): delta method,
 *   m = max(log_w); u = exp(log_w - m); var(u) / (N * mean(u)^2), population variance (ddof=0),
 *   NaN when mean == 0.  np.var = mean(abs(u - mean(u))**2). */
double orc_log_evidence_ratio_variance(int64_t n, const double* ll, const double* lp,
                                       const double* lq, double beta0, double beta) {
    double* lw = (double*)malloc(sizeof(double) * (size_t)n);
    orc_unnormalized_log_weights(n, ll, lp, lq, beta0, beta, lw);
    double m = lw[0];
    int has_nan = isnan(m);
    for (int64_t i = 1; i < n; i++) {
        if (isnan(lw[i])) has_nan = 1;
        if (lw[i] > m) m = lw[i];
    }
    if (has_nan) m = NAN;
    for (int64_t i = 0; i < n; i++) lw[i] = exp(lw[i] - m);
    double mean = pairwise_sum(lw, n) / (double)n;
    for (int64_t i = 0; i < n; i++) {
        double dlt = lw[i] - mean;
        lw[i] = dlt * dlt;
    }
    double var = pairwise_sum(lw, n) / (double)n;
    free(lw);
    if (mean != 0.0) return var / ((double)n * (mean * mean));
    return NAN;
}

/* smc/base.py:114-121  current_target_efficiency(beta) */
double orc_current_target_efficiency(double beta, int adaptive_target, double t0, double t1,
                                     double rate) {
    if (adaptive_target) return t0 + (t1 - t0) * pow(beta, rate);
    return t0;
}

/* smc/base.py:123-213  SMCSampler.determine_beta — the numeric part.
 *   adaptive == 0: beta += beta_step; clamp >= 1 -> 1                      (:162-165)
 *   adaptive == 1: ESS(1.0)/N >= target -> beta_min = 1                    (:170-175)
 *                  bisection beta_try = 0.5*(beta_max+beta_min)            (:177-185)
 *                  optional adaptive min step                              (:198-201)
 *                  beta = max(beta*, prev+min_step); min(beta, prev+max_step, 1) (:202-203)
 *                  unchanged -> BetaScheduleError                          (:204-212)
 * out[0]=beta, out[1]=min_beta_step, out[2]=beta_star, out[3]=#ESS evaluations. */
int orc_determine_beta(int64_t n, const double* ll, const double* lp, const double* lq,
                       double beta, double beta_step, double min_beta_step, double max_beta_step,
                       double beta_tolerance, int adaptive, int adaptive_min_beta_step,
                       int adaptive_target, double t0, double t1, double rate, double* out) {
    int n_eval = 0;
    double beta_star = beta;
    if (!adaptive) {
        beta += beta_step;
        if (beta >= 1.0) beta = 1.0;
        beta_star = beta;
    } else {
        int st = ORC_OK;
        double beta_prev = beta;
        double beta_min = beta_prev;
        double beta_max = 1.0;
        double eff_beta_max = orc_ess_at_beta(n, ll, lp, lq, beta_prev, beta_max, &st) / (double)n;
        n_eval++;
        if (st != ORC_OK) return st;
        double current_eff = orc_current_target_efficiency(beta_prev, adaptive_target, t0, t1, rate);
        if (eff_beta_max >= current_eff) beta_min = 1.0;
        double target_eff = current_eff;
        while (beta_max - beta_min > beta_tolerance) {
            double beta_try = 0.5 * (beta_max + beta_min);
            double eff = orc_ess_at_beta(n, ll, lp, lq, beta_prev, beta_try, &st) / (double)n;
            n_eval++;
            if (st != ORC_OK) return st;
            if (eff >= target_eff)
                beta_min = beta_try;
            else
                beta_max = beta_try;
        }
        beta_star = beta_min;
        if (adaptive_min_beta_step)
            min_beta_step = min_beta_step * (1 - beta_prev) / (1 - beta_star);
        beta = fmax(beta_star, beta_prev + min_beta_step);
        beta = fmin(fmin(beta, beta_prev + max_beta_step), 1.0);
        if (beta == beta_prev) {
            out[0] = beta; out[1] = min_beta_step; out[2] = beta_star; out[3] = (double)n_eval;
            return ORC_ERR_BETA_STALL;
        }
    }
    out[0] = beta;
    out[1] = min_beta_step;
    out[2] = beta_star;
    out[3] = (double)n_eval;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * PCG64 (numpy.random.PCG64 = PCG XSL-RR 128/64, setseq) — numpy is a dependency of the reference
 * (pyproject.toml:19); `rng.choice` at samples.py:1278 draws `random(n)` doubles from it.
 * Published algorithm (O'Neill, pcg-random.org; numpy/random/src/pcg64/pcg64.h):
 *   state = state * MULT + inc (mod 2^128);  out = rotr64(hi ^ lo, hi >> 58) of the NEW state;
 *   random() = (out >> 11) * 2^-53.
 * State layout here: {state_hi, state_lo, inc_hi, inc_lo}, matching
 * `rng.bit_generator.state["state"]` ("state", "inc" as 128-bit ints).
 * ---------------------------------------------------------------------------------------- */
typedef unsigned __int128 u128;
#define PCG_MULT_HI 0x2360ED051FC65DA4ULL
#define PCG_MULT_LO 0x4385DF649FCCF645ULL

static inline u128 mk128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

static inline uint64_t pcg_next64(u128* state, u128 inc) {
    *state = *state * mk128(PCG_MULT_HI, PCG_MULT_LO) + inc;
    uint64_t hi = (uint64_t)(*state >> 64), lo = (uint64_t)*state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63));
}

void orc_pcg64_advance(uint64_t st[4], uint64_t delta_hi, uint64_t delta_lo) {
    u128 state = mk128(st[0], st[1]), inc = mk128(st[2], st[3]);
    u128 delta = mk128(delta_hi, delta_lo);
    u128 cur_mult = mk128(PCG_MULT_HI, PCG_MULT_LO), cur_plus = inc, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    state = acc_mult * state + acc_plus;
    st[0] = (uint64_t)(state >> 64);
    st[1] = (uint64_t)state;
}

/* n doubles of Generator.random(); advances st by n draws. */
void orc_pcg64_random(uint64_t st[4], int64_t n, double* out) {
    u128 state = mk128(st[0], st[1]), inc = mk128(st[2], st[3]);
    for (int64_t i = 0; i < n; i++) out[i] = (double)(pcg_next64(&state, inc) >> 11) * (1.0 / 9007199254740992.0);
    st[0] = (uint64_t)(state >> 64);
    st[1] = (uint64_t)state;
}

/* numpy Generator.choice(N, size=n_out, replace=True, p=w) (numpy/random/_generator.pyx, the
 * `p is not None and replace` branch), as pinned by SURVEY.md F3:
 *   cdf = p.cumsum(); cdf /= cdf[-1]; u = random(n_out); idx = cdf.searchsorted(u, side="right")
 * cumsum is a strictly sequential fp64 accumulation.  `cdf` (n doubles) is caller scratch and
 * holds the normalised cdf on return. */
void orc_cdf_from_weights(int64_t n, const double* w, double* cdf) {
    double s = 0.0;
    if (n > 0) { s = w[0]; cdf[0] = s; }
    for (int64_t i = 1; i < n; i++) {
        s = s + w[i];
        cdf[i] = s;
    }
    double last = cdf[n - 1];
    for (int64_t i = 0; i < n; i++) cdf[i] = cdf[i] / last;
}

void orc_searchsorted_right(int64_t n, const double* cdf, int64_t n_out, const double* u, int64_t* idx) {
    for (int64_t j = 0; j < n_out; j++) {
        int64_t lo = 0, hi = n;
        double key = u[j];
        while (lo < hi) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (cdf[mid] <= key) lo = mid + 1; else hi = mid;
        }
        idx[j] = lo;
    }
}

/* samples.py:1276-1278  the index part of SMCSamples.resample:
 *   log_w = self.log_weights(beta);  w = exp(log_w - logsumexp(log_w));
 *   idx = rng.choice(N, size=n_out, replace=True, p=w)
 * `u` are the n_out uniforms (from orc_pcg64_random or any generator's random(n_out)).
 * uniform_weights != 0 restates the `beta == self.beta, n_samples != N` branch (:1273-1274,
 * log_w = zeros).  numpy's p-validation (sum to 1 within sqrt(eps)) -> ORC_ERR_PSUM. */
int orc_resample_indices(int64_t n, const double* ll, const double* lp, const double* lq,
                         double beta0, double beta, int uniform_weights, int64_t n_out,
                         const double* u, int64_t* idx, double* cdf_out /* n doubles or NULL */) {
    double* lw = (double*)malloc(sizeof(double) * (size_t)n * 3);
    double* tmp = lw + n;
    double* cdf = lw + 2 * n;
    if (uniform_weights) {
        for (int64_t i = 0; i < n; i++) lw[i] = 0.0;
    } else {
        int st = orc_log_weights(n, ll, lp, lq, beta0, beta, lw);
        if (st != ORC_OK) { free(lw); return st; }
    }
    double lse = logsumexp_tmp(lw, n, tmp);
    for (int64_t i = 0; i < n; i++) lw[i] = exp(lw[i] - lse);
    /* numpy: kahan_sum(p) must be within atol=sqrt(eps) of 1 */
    double psum = pairwise_sum(lw, n);
    if (!(fabs(psum - 1.0) <= 1.4901161193847656e-08)) { free(lw); return ORC_ERR_PSUM; }
    orc_cdf_from_weights(n, lw, cdf);
    orc_searchsorted_right(n, cdf, n_out, u, idx);
    if (cdf_out) memcpy(cdf_out, cdf, sizeof(double) * (size_t)n);
    free(lw);
    return ORC_OK;
}

/* Normalised weights only (samples.py:1277) — used by tests to pin w itself. */
int orc_normalized_weights(int64_t n, const double* ll, const double* lp, const double* lq,
                           double beta0, double beta, double* w) {
    double* tmp = (double*)malloc(sizeof(double) * (size_t)n);
    int st = orc_log_weights(n, ll, lp, lq, beta0, beta, w);
    if (st != ORC_OK) { free(tmp); return st; }
    double lse = logsumexp_tmp(w, n, tmp);
    for (int64_t i = 0; i < n; i++) w[i] = exp(w[i] - lse);
    free(tmp);
    return ORC_OK;
}

/* samples.py:1279-1287  x[idx], log_likelihood[idx], log_prior[idx], log_q[idx]  (row gather).
 * elem = bytes per x element (8 fp64 / 4 fp32). */
void orc_gather_rows(int64_t n_out, const int64_t* idx, int d, int elem, const void* x_in, void* x_out,
                     const double* ll_in, const double* lp_in, const double* lq_in, double* ll_out,
                     double* lp_out, double* lq_out) {
    size_t row = (size_t)d * (size_t)elem;
    for (int64_t j = 0; j < n_out; j++) {
        memcpy((char*)x_out + (size_t)j * row, (const char*)x_in + (size_t)idx[j] * row, row);
        ll_out[j] = ll_in[idx[j]];
        lp_out[j] = lp_in[idx[j]];
        lq_out[j] = lq_in[idx[j]];
    }
}

/* Systematic / stratified uniforms — NO reference counterpart (the reference
 * resampler is multinomial).  Contract defined by this repository:
 *   systematic: u_j = (j + u0) / n_out with one uniform u0;  stratified: u_j = (j + v_j) / n_out. */
void orc_systematic_uniforms(int64_t n_out, double u0, double* u) {
    for (int64_t j = 0; j < n_out; j++) u[j] = ((double)j + u0) / (double)n_out;
}
void orc_stratified_uniforms(int64_t n_out, const double* v, double* u) {
    for (int64_t j = 0; j < n_out; j++) u[j] = ((double)j + v[j]) / (double)n_out;
}

/* samples.py:1217-1219 + smc/base.py:507-519  tempered log-target used by the mutation step:
 *   log_p_t = (1 - beta) * log_q + beta * (log_likelihood + log_prior) [+ log|det J|]; NaN -> -inf */
double orc_log_p_t(double ll, double lp, double lq, double beta) {
    double r = (1 - beta) * lq + beta * (ll + lp);
    if (isnan(r)) r = -INFINITY;
    return r;
}

/* mcmc.py:88-90,107-108  draw_initial_samples validity filter: keep rows with finite log_prior and
 * finite log_likelihood, in order; returns the number kept (rows compacted in place to the front
 * of the *_out arrays). */
int64_t orc_compact_valid(int64_t n, int d, const double* x, const double* ll, const double* lp,
                          const double* lq, double* x_out, double* ll_out, double* lp_out,
                          double* lq_out) {
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        if (isfinite(lp[i]) && isfinite(ll[i])) {
            memcpy(x_out + (size_t)k * d, x + (size_t)i * d, sizeof(double) * (size_t)d);
            ll_out[k] = ll[i]; lp_out[k] = lp[i]; lq_out[k] = lq[i];
            k++;
        }
    }
    return k;
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 counter RNG (Salmon et al., SC'11 "Parallel random numbers: as easy as 1, 2, 3";
 * Random123 v1.14 constants) — the build's in-kernel generator for the pCN noise.  Known-answer
 * vectors from Random123's kat_vectors are checked in tests/test_oracle.py.
 * ---------------------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Two 32-bit words -> uniform double in (0,1): ((hi<<21 | lo>>11) + 0.5) * 2^-53 (53 random bits,
 * never 0 or 1).  DESIGN.md §RNG. */
static inline double u01_from_words(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 21) ^ ((uint64_t)lo >> 11);
    v &= ((1ULL << 53) - 1);
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

/* One Philox block -> two standard normals by Box-Muller:
 *   r = sqrt(-2 log u1); (z0, z1) = r * (cos(2 pi u2), sin(2 pi u2)). */
static inline void normal_pair(const uint32_t w[4], double* z0, double* z1) {
    double u1 = u01_from_words(w[0], w[1]);
    double u2 = u01_from_words(w[2], w[3]);
    double r = sqrt(-2.0 * log(u1));
    double a = 6.283185307179586476925286766559 * u2;
    *z0 = r * cos(a);
    *z1 = r * sin(a);
}

/* Noise for particle `gid` at Markov step `step`: xi[0..d) ~ N(0,1), plus the accept uniform.
 * Counter layout: ctr = {gid_lo, gid_hi, step, slot}; slot p<d/2 -> normals
 * (2p, 2p+1); slot 0xFFFFFFFF -> accept uniform (words 0,1).  key = {seed_lo, seed_hi}. */
void orc_pcn_noise(uint64_t seed, uint64_t gid, uint32_t step, int d, double* xi, double* u_acc) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), step, 0};
    uint32_t w[4];
    for (int p = 0; 2 * p < d; p++) {
        ctr[3] = (uint32_t)p;
        orc_philox4x32_10(ctr, key, w);
        double z0, z1;
        normal_pair(w, &z0, &z1);
        xi[2 * p] = z0;
        if (2 * p + 1 < d) xi[2 * p + 1] = z1;
    }
    ctr[3] = 0xFFFFFFFFu;
    orc_philox4x32_10(ctr, key, w);
    *u_acc = u01_from_words(w[0], w[1]);
}

/* Fast noise mode (ASMC_NOISE_F32): one Philox block -> four normals by fp32 Box-Muller.
 * Counter slot = quad index | 0x40000000.  The GPU uses the hardware transcendental units, so parity with
 * this libm restatement is ~1e-6 relative, not bitwise. */
void orc_pcn_noise_f32(uint64_t seed, uint64_t gid, uint32_t step, int d, double* xi, double* u_acc) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), step, 0};
    uint32_t w[4];
    const float k = 2.3283064365386963e-10f, h = 1.1641532182693481e-10f;
    for (int q = 0; 4 * q < d; q++) {
        ctr[3] = (uint32_t)q | 0x40000000u;
        orc_philox4x32_10(ctr, key, w);
        float u0 = fmaf((float)w[0], k, h), u1 = fmaf((float)w[2], k, h);
        float t0 = (float)w[1] * k, t1 = (float)w[3] * k;
        float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u1));
        float z[4] = {r0 * cosf(6.28318530717958647692f * t0), r0 * sinf(6.28318530717958647692f * t0),
                      r1 * cosf(6.28318530717958647692f * t1), r1 * sinf(6.28318530717958647692f * t1)};
        for (int e = 0; e < 4 && 4 * q + e < d; e++) xi[4 * q + e] = (double)z[e];
    }
    ctr[3] = 0xFFFFFFFFu;
    orc_philox4x32_10(ctr, key, w);
    *u_acc = u01_from_words(w[0], w[1]);
}

/* Built-in device targets: each of log_likelihood / log_prior / log_q is a
 * diagonal-Gaussian-mixture log-density
 *     log sum_c exp( logw_c - 0.5 * sum_j (x_j - mu_cj)^2 * prec_cj )          (C components)
 * with logw_c already containing the normalisation constants.  C == 1 reduces to
 * logw_0 - 0.5 * sum_j (x_j - mu_j)^2 * prec_j with no exp/log.  Sum over j in index order. */
double orc_diag_mixture_logpdf(int d, int C, const double* logw, const double* mu, const double* prec,
                               const double* x) {
    double best = -INFINITY;
    double terms[16];
    if (C > 16) C = 16;
    for (int c = 0; c < C; c++) {
        double q = 0.0;
        for (int j = 0; j < d; j++) {
            double t = x[j] - mu[(size_t)c * d + j];
            q += t * t * prec[(size_t)c * d + j];
        }
        terms[c] = logw[c] - 0.5 * q;
        if (terms[c] > best) best = terms[c];
    }
    if (C == 1) return terms[0];
    if (best == -INFINITY) return -INFINITY;
    double s = 0.0;
    for (int c = 0; c < C; c++) s += exp(terms[c] - best);
    return best + log(s);
}

/* One pCN step for the whole population — THIS REPOSITORY's specification (parity unpinned vs
 * minipcn; interface pinned by reference tests/test_samplers/test_mcmc/test_checkpointing.py:88-103):
 *   y  = Linv (x - mu)                 (whitened coordinates, L = chol(cov), Linv lower-triangular)
 *   y' = sqrt(1 - rho^2) y + rho xi,   x' = mu + L y'
 *   log a = [log p_t(x') + 0.5 |y'|^2] - [log p_t(x) + 0.5 |y|^2]
 *   accept iff log(u) < log a.
 * Targets are three diag-mixtures (ll, lp, lq).  Arrays updated in place; returns #accepted.
 * tgt layout per density: C, then logw[C], mu[C*d], prec[C*d] passed as separate pointers. */
typedef struct {
    int C;
    const double* logw;
    const double* mu;
    const double* prec;
} orc_mixture;

int64_t orc_pcn_step(int64_t n, int d, double* x, double* ll, double* lp, double* lq, double beta,
                     const double* mu, const double* L, const double* Linv, double rho,
                     const orc_mixture* t_ll, const orc_mixture* t_lp, const orc_mixture* t_lq,
                     uint64_t seed, uint64_t gid0, uint32_t step, int noise_f32) {
    int64_t n_acc = 0;
    double* buf = (double*)malloc(sizeof(double) * (size_t)d * 4);
    double *y = buf, *yp = buf + d, *xp = buf + 2 * d, *xi = buf + 3 * d;
    const double a = sqrt(1.0 - rho * rho);
    for (int64_t i = 0; i < n; i++) {
        double* xr = x + (size_t)i * d;
        double u;
        if (noise_f32) orc_pcn_noise_f32(seed, gid0 + (uint64_t)i, step, d, xi, &u);
        else orc_pcn_noise(seed, gid0 + (uint64_t)i, step, d, xi, &u);
        double q0 = 0.0, q1 = 0.0;
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int k = 0; k <= j; k++) s += Linv[(size_t)j * d + k] * (xr[k] - mu[k]);
            y[j] = s;
            q0 += s * s;
        }
        for (int j = 0; j < d; j++) {
            yp[j] = a * y[j] + rho * xi[j];
            q1 += yp[j] * yp[j];
        }
        for (int j = 0; j < d; j++) {
            double s = 0.0;
            for (int k = 0; k <= j; k++) s += L[(size_t)j * d + k] * yp[k];
            xp[j] = mu[j] + s;
        }
        double nll = orc_diag_mixture_logpdf(d, t_ll->C, t_ll->logw, t_ll->mu, t_ll->prec, xp);
        double nlp = orc_diag_mixture_logpdf(d, t_lp->C, t_lp->logw, t_lp->mu, t_lp->prec, xp);
        double nlq = orc_diag_mixture_logpdf(d, t_lq->C, t_lq->logw, t_lq->mu, t_lq->prec, xp);
        double lp_new = orc_log_p_t(nll, nlp, nlq, beta);
        double lp_old = orc_log_p_t(ll[i], lp[i], lq[i], beta);
        double log_a = (lp_new + 0.5 * q1) - (lp_old + 0.5 * q0);
        if (log(u) < log_a) {
            memcpy(xr, xp, sizeof(double) * (size_t)d);
            ll[i] = nll; lp[i] = nlp; lq[i] = nlq;
            n_acc++;
        }
    }
    free(buf);
    return n_acc;
}

/* Step-size adaptation (this repository's spec):
 *   log rho <- log rho + (acc - target) / (t + 1)^0.75,  rho clipped to [1e-4, 0.99]. */
double orc_pcn_adapt(double rho, double acc, double target, int t) {
    double lr = log(rho) + (acc - target) / pow((double)(t + 1), 0.75);
    double r = exp(lr);
    if (r < 1e-4) r = 1e-4;
    if (r > 0.99) r = 0.99;
    return r;
}

/* Column means and covariance (ddof = 1, like numpy.cov) of x [n,d]: two-pass. */
void orc_moments(int64_t n, int d, const double* x, double* mean, double* cov) {
    for (int j = 0; j < d; j++) mean[j] = 0.0;
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < d; j++) mean[j] += x[(size_t)i * d + j];
    for (int j = 0; j < d; j++) mean[j] /= (double)n;
    for (int j = 0; j < d * d; j++) cov[j] = 0.0;
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < d; j++) {
            double dj = x[(size_t)i * d + j] - mean[j];
            for (int k = 0; k <= j; k++) cov[(size_t)j * d + k] += dj * (x[(size_t)i * d + k] - mean[k]);
        }
    for (int j = 0; j < d; j++)
        for (int k = 0; k <= j; k++) {
            cov[(size_t)j * d + k] /= (double)(n - 1);
            cov[(size_t)k * d + j] = cov[(size_t)j * d + k];
        }
}

/* One full SMC temperature iteration of the IS-only path (a3-a9: determine_beta, ESS(beta),
 * ESS(1), evidence ratio + variance, resample indices, gather) — used as the bounded CPU baseline
 * in bench.py (kind "port").  Returns the new beta; outputs in *_out. */
int orc_is_iteration(int64_t n, int d, const double* x, const double* ll, const double* lp,
                     const double* lq, double beta0, double target_eff, double tol, uint64_t rng[4],
                     double* x_out, double* ll_out, double* lp_out, double* lq_out, double* scalars /*6*/) {
    double out[4];
    int st = orc_determine_beta(n, ll, lp, lq, beta0, NAN, 0.0, 1.0, tol, 1, 0, 0, target_eff, 0, 1, out);
    if (st != ORC_OK) return st;
    double beta = out[0];
    scalars[0] = beta;
    scalars[1] = orc_ess_at_beta(n, ll, lp, lq, beta0, beta, &st);
    scalars[2] = orc_ess_at_beta(n, ll, lp, lq, beta0, 1.0, &st);
    scalars[3] = orc_log_evidence_ratio(n, ll, lp, lq, beta0, beta);
    scalars[4] = orc_log_evidence_ratio_variance(n, ll, lp, lq, beta0, beta);
    scalars[5] = out[3];
    double* u = (double*)malloc(sizeof(double) * (size_t)n);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    orc_pcg64_random(rng, n, u);
    st = orc_resample_indices(n, ll, lp, lq, beta0, beta, 0, n, u, idx, NULL);
    if (st == ORC_OK) orc_gather_rows(n, idx, d, 8, x, x_out, ll, lp, lq, ll_out, lp_out, lq_out);
    free(u);
    free(idx);
    return st;
}

/* ------------------------------------------------------------------------------------------
 * Coupling-flow log-density, fp32 (the flow's dtype; reference flows default to float32,
 * src/aspire/flows/torch/flows.py:31-35).  Same role as ZukoFlow.log_prob
 * (src/aspire/flows/torch/flows.py:368-387): standardise, run the flow data -> latent, return
 * base log-density + log|det J|.  The flow is this repository's RealNVP (aspire_amd/flows.py):
 * masks alternate first half / second half; conditioner MLP d/2 -> hidden -> hidden -> d with ReLU
 * (torch.nn.Linear layout [out, in]); s = 2 tanh(s_raw/2); z_b = (x_b - t) exp(-s).
 * weights[3c+k] / biases[3c+k]: layer k of coupling layer c.  x is [n, d] fp64 and is cast to fp32
 * first (torch.as_tensor(x, dtype=float32)).  Dot products are accumulated in index order.
 * ---------------------------------------------------------------------------------------- */
static void dense_f32(int n_out, int n_in, const float* W, const float* b, const float* in, float* out,
                      int relu) {
    for (int o = 0; o < n_out; o++) {
        float acc = b[o];
        for (int k = 0; k < n_in; k++) acc = fmaf(W[(int64_t)o * n_in + k], in[k], acc);
        out[o] = (relu && acc < 0.0f) ? 0.0f : acc;
    }
}

int orc_coupling_logprob(int64_t n, int d, const double* x, int n_layers, int hidden,
                         const float* const* weights, const float* const* biases, const float* loc,
                         const float* scale, double* out) {
    if (d < 2 || d % 2 || n_layers < 1 || hidden < 1) return ORC_ERR_ARG;
    const int dh = d / 2;
    float* z = (float*)malloc(sizeof(float) * (size_t)(d + 2 * hidden + d));
    float *h1 = z + d, *h2 = h1 + hidden, *o = h2 + hidden;
    float log_scale_sum = 0.0f;
    for (int j = 0; j < d; j++) log_scale_sum += logf(scale[j]);
    for (int64_t i = 0; i < n; i++) {
        for (int j = 0; j < d; j++) z[j] = ((float)x[i * d + j] - loc[j]) / scale[j];
        float ladj = -log_scale_sum;
        for (int c = 0; c < n_layers; c++) {
            float* cond = (c % 2 == 0) ? z : z + dh;
            float* trans = (c % 2 == 0) ? z + dh : z;
            dense_f32(hidden, dh, weights[3 * c], biases[3 * c], cond, h1, 1);
            dense_f32(hidden, hidden, weights[3 * c + 1], biases[3 * c + 1], h1, h2, 1);
            dense_f32(d, hidden, weights[3 * c + 2], biases[3 * c + 2], h2, o, 0);
            float ssum = 0.0f;
            for (int j = 0; j < dh; j++) {
                const float sj = 2.0f * tanhf(o[j] * 0.5f);
                trans[j] = (trans[j] - o[dh + j]) * expf(-sj);
                ssum += sj;
            }
            ladj -= ssum;
        }
        float q = 0.0f;
        for (int j = 0; j < d; j++) q += z[j] * z[j];
        out[i] = (double)((-0.5f * q - 0.5f * (float)d * 1.8378770664093453f) + ladj);
    }
    free(z);
    return ORC_OK;
}
