/*
 * asmc.h — C ABI of libasmc_hip.so: aspire's SMC particle-batch hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference (mj-will/aspire) is pure Python;
 * the interfaces replaced are numpy/array-API expressions inside
 *   src/aspire/samples.py          SMCSamples (weights, evidence, resample)
 *   src/aspire/utils.py            logsumexp, effective_sample_size
 *   src/aspire/samplers/smc/base.py  determine_beta / sample loop / log_prob
 *   src/aspire/samplers/smc/minipcn.py  mutate (third-party minipcn kernel)
 *   src/aspire/samplers/mcmc.py    draw_initial_samples
 * Each entry point below cites the reference file:line (relative to the reference root) it
 * replaces.  A maintainer binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns int: 0 = ASMC_OK, <0 = error (asmc_last_error() has the text);
 *     no C++ exception crosses the ABI; NaN detection is returned as a count so the host can
 *     raise the reference's exception types.
 *   - pointers named *_dev are device (HBM) pointers owned by the caller; *_host are host
 *     pointers.  The library never frees caller memory and allocates nothing after
 *     asmc_ctx_create (scratch lives in the ctx).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls that return
 *     host scalars synchronise that stream; all others only enqueue.
 *   - particle state layout: x row-major [N, d] (particle-major; fp64 or fp32, `x_dtype`),
 *     log_likelihood / log_prior / log_q as three fp64 vectors [N].
 *   - a ctx is bound to one device and one host thread.
 */
#ifndef ASMC_H
#define ASMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASMC_ABI_VERSION 23

#define ASMC_OK 0
#define ASMC_ERR_ARG (-1)
#define ASMC_ERR_HIP (-2)
#define ASMC_ERR_NOMEM (-3)
#define ASMC_ERR_UNSUPPORTED (-4)

#define ASMC_F64 0
#define ASMC_F32 1

#define ASMC_MAX_BETAS 32      /* candidate betas evaluated per pass */
#define ASMC_BIS_REC 40        /* doubles per rank record of the sharded beta search (asmc_find_beta_shard_*) */
#define ASMC_STUDENT_MAX_ROWS 16384 /* largest subsample of the tpCN reference fit (asmc_student_*) */
#define ASMC_CDF_REC 9         /* int64 words per tile record of the sharded exact cdf (asmc_cdf_shard_*) */
#define ASMC_CDF_STATE 36      /* doubles per rank state of its chain (asmc_cdf_shard_chain) */
#define ASMC_SELECT_THREADS 262144 /* generator threads of asmc_pcg64_select (fixes the order of the kept draws) */
#define ASMC_MAX_COMPONENTS 8  /* mixture components of a built-in density */
#define ASMC_MAX_DIMS 256

#define ASMC_NOISE_F64 0
#define ASMC_NOISE_F32 1

#define ASMC_CDF_EXACT 0 /* sequential-order fp64 rounding == numpy cumsum (bit-exact) */
#define ASMC_CDF_FAST 1  /* parallel-order rounding */
#define ASMC_CDF_NORMALIZE 0x100 /* OR into the mode: write cdf / cdf[-1] (numpy's `cdf /= cdf[-1]`) in the same pass */

typedef struct asmc_ctx asmc_ctx;
typedef void* asmc_stream;

/* Built-in log-density: diagonal Gaussian mixture
 *   log sum_c exp(logw[c] - 0.5 * sum_j (x_j - mu[c,j])^2 * prec[c,j]),  logw includes constants.
 * All pointers are DEVICE pointers (fp64). */
typedef struct {
    int32_t n_components;
    int32_t reserved;
    const double* logw_dev; /* [C] */
    const double* mu_dev;   /* [C, d] */
    const double* prec_dev; /* [C, d] */
} asmc_mixture;

/* Parameters of the fused pCN mutation (this repository's pCN specification, DESIGN.md §pCN;
 * replaces minipcn.Sampler(...).sample(z, n_steps) at reference samplers/smc/minipcn.py:97-114
 * and the tempered target of samplers/smc/base.py:507-519 + samples.py:1217-1219). */
typedef struct {
    int32_t d;
    int32_t x_dtype;          /* ASMC_F64 | ASMC_F32 */
    double beta;              /* inverse temperature of the tempered target */
    const double* mu_dev;     /* [d] reference-Gaussian mean */
    const double* L_dev;      /* [d,d] row-major lower Cholesky factor of the covariance */
    const double* Linv_dev;   /* [d,d] row-major inverse of L */
    asmc_mixture log_likelihood;
    asmc_mixture log_prior;
    asmc_mixture log_q;
    uint64_t seed;            /* Philox key */
    uint64_t gid0;            /* global index of local particle 0 (sharded runs) */
    double target_accept;     /* e.g. 0.234 (reference minipcn.py:47) */
    int32_t adapt;            /* 0: fixed step size; 1: Robbins-Monro adaptation on the device after every step; k >= 2: the same
                                 updates applied in blocks of k steps (see asmc_pcn_set_count_cells) */
    int32_t noise;            /* ASMC_NOISE_F64: fp64 Box-Muller, 2 normals / Philox block (1e-15 parity with
                                 the oracle); ASMC_NOISE_F32: fp32 hardware Box-Muller, 4 normals / block
                                 (fast; honoured by the register-resident kernels, d in {4,8,16,32}) */
    double nu;                /* > 0 (>= 1): t-preconditioned Crank-Nicolson, the reference's default step_fn="tpcn"
                                 (smc/minipcn.py:46-49): the reference distribution is Student-t(mu, L L^T, nu); the
                                 proposal noise is scaled by sqrt(s), s ~ InvGamma((d + nu)/2, (nu + |y|^2)/2), and the
                                 acceptance ratio carries ((d + nu)/2) log(1 + |y|^2/nu) instead of |y|^2/2.
                                 <= 0: Gaussian reference (plain pCN) */
} asmc_pcn_params;

/* ---- library / context ---------------------------------------------------------------- */
int asmc_abi_version(void);
const char* asmc_last_error(void); /* thread-local */
int asmc_device_count(int* n_out);
int asmc_ctx_create(asmc_ctx** ctx_out, int device, int64_t n_max, int d_max);
int asmc_ctx_destroy(asmc_ctx* ctx);

/* Per-kernel timing with HIP events recorded on the launch stream around every kernel of the library
 * (measurement aid for bench.py's roofline leg; off by default, ~2 event records per launch when on).
 * asmc_profile_report writes lines "<kernel> <launches> <avg_ms>\n" into buf and clears the log. */
int asmc_profile_enable(asmc_ctx* ctx, int on);
int asmc_profile_report(asmc_ctx* ctx, char* buf_host, int64_t buf_len);

/* ---- weighting / ESS / evidence (reference samples.py:1221-1249, utils.py:248-255,510-512) ---
 * Unnormalised tempered log-weight, exactly the reference association (samples.py:1222-1224):
 *     lw_i(beta) = (beta0 - beta) * lq_i + (beta - beta0) * (ll_i + lp_i)
 * evaluated for K candidate betas in one pass over (ll, lp, lq).
 *
 * asmc_weights_max : m_k = max_i lw_i(beta_k); n_nan = #NaN log-weights over all k (reference
 *                    raises ValueError when > 0, samples.py:1246-1247).
 * asmc_weights_sums: S1_k = sum_i exp(t), S2_k = sum_i exp(t)^2, t = (lw_i(beta_k) + shift_k) - m_k.
 *                    out_host = {S1_0, S2_0, S1_1, S2_1, ...}.  shift_host may be NULL (zeros).
 * asmc_weights_stats: both passes back-to-back, out_host[k*4..] = {m, S1, S2, n_nan}.
 * From these the host forms logsumexp = m + log S1 (utils.py:248-255), the ESS
 * exp(2*LSE(lw) - LSE(2 lw)) (utils.py:510-512) and the evidence ratio (samples.py:1226-1228).
 * asmc_weights_m2  : sum_i (exp(lw_i - m) - mean_u)^2, the numerator of the population variance in
 *                    log_evidence_ratio_variance (samples.py:1230-1242). */
int asmc_weights_max(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                     const double* lq_dev, double beta0, const double* betas_host, int K,
                     double* m_host, int64_t* n_nan_host, asmc_stream stream);
int asmc_weights_sums(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                      const double* lq_dev, double beta0, const double* betas_host,
                      const double* m_host, const double* shift_host, int K, double* out_host,
                      asmc_stream stream);
int asmc_weights_stats(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                       const double* lq_dev, double beta0, const double* betas_host, int K,
                       double* out_host, asmc_stream stream);
/* SMCSampler.determine_beta's adaptive search (smc/base.py:167-186) without host round trips: ESS(1.0) check,
 * then k-ary bisection rounds (15 midpoints = 4 levels per round, the reference's 0.5*(max+min) values) chained
 * on the stream, two launches per round; decisions use the same scalar formulas on device.  The round-0
 * pass (beta = 1) finds the exact maximum; later rounds shift their log-sum-exps by
 * m(1) (beta - beta0) / (1 - beta0), which equals the maximum up to rounding (lw is linear in beta).
 * Single rank; asmc_find_beta_shard_* below is the sharded form.
 * out_host[13] = {beta_star (= beta_min), beta_max, converged, device rounds, ESS(1.0)/N, n_nan,
 *                m, S1, S2 of the log-sum-exp at beta_star, 1 if that triple is valid, m, S1, S2 at beta = 1}. */
int asmc_find_beta(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                   const double* lq_dev, double beta0, double target_eff, double tol,
                   double* out_host, asmc_stream stream);

/* One iteration's importance step for a single-rank population, enqueued without a host round trip
 * (smc/base.py:167-186 determine_beta, samples.py:1226-1249 evidence ratio / variance, samples.py:1276-1278
 * resampling indices): asmc_find_beta + asmc_weights_m2_lse + asmc_normalized_weights + asmc_cdf(EXACT | NORMALIZE) +
 * asmc_pcg64_uniforms + asmc_search as five launches (persistent weight kernel; transducer, chain and write passes of
 * the exact scan, the write pass filling the search's guide table; search with the uniforms generated in registers).
 * idx_out[j] (device, n_out entries) = searchsorted(cumsum(w) / cumsum(w)[-1], u_j, side="right") for the next n_out
 * doubles u of the PCG64 stream `rng_state` = {state_hi, state_lo, inc_hi, inc_lo}; the caller advances its generator by
 * n_out once it accepts the step.  w_scratch, cdf_scratch: n doubles each (device).
 * SPECULATIVE: beta* is only known on the device.  When the search finds no valid beta* (NaN log-weights, no candidate
 * above beta0, not converged) the weights are uniform (idx_out is then a valid but meaningless draw) and
 * asmc_importance_result reports found = 0; the caller redoes the step through the step-by-step entry points.
 * asmc_importance_result (synchronises): out_host[16] = asmc_find_beta's 13 values, then
 * [13] sum (exp(lw - m) - mean_u)^2, [14] S1' of the second log-sum-exp (both at beta*), [15] found. */
int asmc_importance_step(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev, const double* lq_dev,
                         double beta0, double target_eff, double tol, const uint64_t rng_state[4], int64_t n_out,
                         double* w_scratch_dev, double* cdf_scratch_dev, int64_t* idx_out_dev, asmc_stream stream);
int asmc_importance_result(asmc_ctx* ctx, double* out_host, asmc_stream stream);
/* Puts asmc_importance_result's read-back on the stream now, with an event behind it: the next asmc_importance_result waits
 * for that event only - not for what the caller has enqueued behind the step since (asmc_mean_gram_enqueue on the gathered
 * rows) - so the host can go on with the schedule while those passes run. */
int asmc_importance_result_enqueue(asmc_ctx* ctx, asmc_stream stream);
/* 1 while asmc_importance_step can be used on this ctx; 0 once a launch of its persistent kernel timed out at a grid barrier
 * (not fully resident, e.g. another process's kernel of the same kind on the GPU): that step reported found = 0, the counters
 * were reset, and the caller stays on the step-by-step entry points. */
int asmc_importance_available(asmc_ctx* ctx);
/* The same search with the particles sharded over `world` ranks (one process per GPU; the reference has no
 * distributed mode).  A round is split at the rank boundary and never synchronises with the host:
 *   asmc_find_beta_shard_reduce  this rank's sums of the round's 16 candidates -> rec_dev[ASMC_BIS_REC]
 *                                ([0..31] column sums, [32] the shift base they are relative to, [33] NaN log-weights);
 *   (caller)                     all-gather of the records in rank order (RCCL all_gather_into_tensor on the stream);
 *   asmc_find_beta_shard_decide  every rank merges recs_dev[world][ASMC_BIS_REC] in rank order, takes the round's four
 *                                bisection decisions and writes the next candidates (identical on every rank).
 * Rounds 0, 1, 2, ... until asmc_find_beta_shard_result (the only synchronising call; same out_host[13] as
 * asmc_find_beta, N = n_global) reports convergence; rounds after convergence are no-ops.  In round 0 the ranks
 * shift by their local maximum at beta = 1 and the decide step rescales to the global one, so no separate
 * all-reduce(max) is needed. */
int asmc_find_beta_shard_reduce(asmc_ctx* ctx, int64_t n_local, const double* ll_dev, const double* lp_dev,
                                const double* lq_dev, double beta0, int round, double* rec_dev, asmc_stream stream);
int asmc_find_beta_shard_decide(asmc_ctx* ctx, const double* recs_dev, int world, int64_t n_global, double beta0,
                                double target_eff, double tol, int round, asmc_stream stream);
/* rounds first .. last-1 in one call: reduce -> ncclAllGather of the records on the library's own communicator (same
 * stream) -> decide; rec_dev[ASMC_BIS_REC], recs_dev[world * ASMC_BIS_REC] are the caller's */
int asmc_find_beta_shard_rounds(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev, const double* lq_dev,
                                double beta0, double target_eff, double tol, int world, int64_t n_global, double* rec_dev,
                                double* recs_dev, int first, int last, asmc_stream stream);
int asmc_find_beta_shard_result(asmc_ctx* ctx, double* out_host, asmc_stream stream);
int asmc_weights_m2(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                    const double* lq_dev, double beta0, double beta, double m, double mean_u,
                    double* m2_host, asmc_stream stream);
/* asmc_weights_m2 and, in the same pass, the sum of the second log-sum-exp of the resampling step
 * (w = exp(log_w - logsumexp(log_w)) over the SHIFTED log-weights, samples.py:1277 on :1244-1249):
 * out_host[2] = { sum (exp(lw - m) - mean_u)^2 ,  sum exp((lw + shift) - mp) }. */
int asmc_weights_m2_lse(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                        const double* lq_dev, double beta0, double beta, double m, double mean_u,
                        double shift, double mp, double* out_host, asmc_stream stream);
/* ... with the two sums left in device memory (out_dev[2]) and no host synchronisation: sharded runs all-gather them
 * together with the local cdf total (asmc_cdf_total_dev). */
int asmc_weights_m2_lse_dev(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                            const double* lq_dev, double beta0, double beta, double m, double mean_u,
                            double shift, double mp, double* out_dev, asmc_stream stream);

/* The sharded importance step without a host decision between its collectives (one process per GPU; the reference has no
 * distributed mode, SURVEY.md §8e - the contract is samples.py:1221-1287 on the GLOBAL population).  The search state that
 * asmc_find_beta_shard_decide leaves on the device (beta*, the (m, S1, S2) triple at beta*, N, log N; identical on every
 * rank) parameterises the passes behind it:
 *   asmc_find_beta_shard_round     a search round WITHOUT a decide launch: round r >= 1 closes round r - 1 itself - every
 *                                  block merges recs_prev_dev[world][ASMC_BIS_REC] (the all-gathered records of round r - 1)
 *                                  in rank order and takes asmc_find_beta_shard_decide's decisions on a block-local copy of
 *                                  the state (fixed order: the same bits in every block and on every rank) - then reduces
 *                                  this rank's sums of round r -> rec_dev.  Round 0 = asmc_find_beta_shard_reduce(round 0).
 *                                  (caller) all-gather of rec_dev after every round;
 *   asmc_weights_m2_lse_shard      closes the LAST round (n_rounds - 1) the same way when recs_last_dev is given (NULL: the
 *                                  state asmc_find_beta_shard_decide left), then asmc_weights_m2_lse_dev with beta, m,
 *                                  mean_u = S1/N, shift = (m + log S1) - log N and
 *                                  mp = m + shift formed on the device in that order -> out_dev[2] (one launch: the block
 *                                  that arrives last adds up the blocks' partials in the two-launch form's order);
 *   (caller)                       all-gather of the pairs in rank order -> parts_dev[world][2];
 *   asmc_normalized_weights_shard  S1' = sum of parts[r][1] in rank order, lse = mp + log S1', w = exp((lw + shift) - lse)
 *                                  -> w_out_dev; carry_out_dev[0] = (sum of the lower ranks' parts[r][1]) / S1', the
 *                                  approximate incoming sum of asmc_cdf_shard_records_dev; tile_sums_dev
 *                                  [asmc_cdf_shard_tiles(n)] = the weights' sums per 2048-particle scan tile (the records
 *                                  pass's prefix hints); state_copy_dev[40] (optional) = the search state, so that it can sit
 *                                  in one buffer with the gathered pairs for asmc_shard_step_result;
 *   asmc_cdf_shard_records_dev     asmc_cdf_shard_records with approx_carry read from the device and, when tile_sums_dev is
 *                                  given, without its own tile-sum and scan launches;
 *   asmc_cdf_shard_chain x 2       as before (the first round zeroes its own scratch and state);
 *   asmc_cdf_shard_finish_select   asmc_cdf_shard_finish + asmc_select_range_dev in three launches: the slice's edges
 *                                  {fail, total, lo, hi} -> edges_out_dev[4], the draws u with lo <= u < hi in index order ->
 *                                  out_dev, {kept, fail} -> info_dev[2];
 *   (caller)                       all-gather of the info pairs;
 *   asmc_shard_step_result         THE synchronisation: res_dev = {state copy [40], parts [2 world], info pairs [2 world]
 *                                  (int64)} in ONE buffer -> out_host[13 + 4 world] = asmc_find_beta_shard_result's 13
 *                                  values, parts, then the (kept, fail) pairs as doubles.
 * Everything in front of asmc_shard_step_result is enqueue-only, so the whole chain can sit behind a mutation's step loop.
 * A search that did not converge within the rounds enqueued (or met NaN weights) leaves uniform weights 1/N and
 * carry_uniform behind - every later launch stays well defined - and the host, seeing converged = 0, discards the chain and
 * continues the search round by round. */
int asmc_find_beta_shard_round(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev, const double* lq_dev,
                               double beta0, double target_eff, double tol, int world, int64_t n_global, int round,
                               const double* recs_prev_dev, double* rec_dev, asmc_stream stream);
int asmc_weights_m2_lse_shard(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev, const double* lq_dev,
                              double* out_dev, const double* recs_last_dev, int world, int64_t n_global, double beta0,
                              double target_eff, double tol, int n_rounds, asmc_stream stream);
int asmc_normalized_weights_shard(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev, const double* lq_dev,
                                  const double* parts_dev, int world, int rank, double carry_uniform, double* w_out_dev,
                                  double* carry_out_dev, double* tile_sums_dev, double* state_copy_dev, int pack_records,
                                  asmc_stream stream);
/* pack_records = 1: the pass also leaves the (ll, lp, lq, 0) records asmc_gather reads per draw in the context (what asmc_gather's
 * own packing pass would write).  asmc_rec_token right after the call names that generation of the records; asmc_rec_claim(token,
 * n, ll, lp, lq) directly in front of the asmc_gather of exactly these arrays makes the gather use them - if no other pass has
 * rewritten the records since (returns 1; 0: the gather packs for itself). */
int64_t asmc_rec_token(asmc_ctx* ctx);
int asmc_rec_claim(asmc_ctx* ctx, int64_t token, int64_t n, const double* ll_dev, const double* lp_dev, const double* lq_dev);
int asmc_shard_step_result(asmc_ctx* ctx, const double* res_dev, int world, double* out_host, asmc_stream stream);
/* asmc_shard_step_result and, when the step it reads back is a finished one (converged, no NaN weights, no slice that needs the
 * replicated scan, every rank's offspring count in (0, cap] - rank-uniform conditions), this rank's asmc_search over its kept
 * draws (kept_dev, as asmc_cdf_shard_finish_select left them) and asmc_gather of its offspring into the caller's cap-row
 * buffers, enqueued from C right behind the synchronisation; *launched = 1 then, the rows written = this rank's count in
 * out_host.  rec_token: asmc_rec_token after asmc_normalized_weights_shard (0: none). */
int asmc_shard_step_finish(asmc_ctx* ctx, const double* res_dev, int world, int rank, int64_t n_local, const double* cdf_dev,
                           const double* kept_dev, int64_t cap, int64_t* idx_dev, int d, int x_dtype, const void* x_in_dev,
                           void* x_out_dev, const double* ll_in_dev, const double* lp_in_dev, const double* lq_in_dev,
                           double* ll_out_dev, double* lp_out_dev, double* lq_out_dev, int64_t rec_token, double* out_host,
                           int* launched, asmc_stream stream);

/* SMCSamples.log_weights(beta) as an array (samples.py:1244-1249): lw_out = lw(beta) + shift,
 * shift = logsumexp(lw) - log N formed on the host from asmc_weights_stats. */
int asmc_log_weights(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                     const double* lq_dev, double beta0, double beta, double shift,
                     double* lw_out_dev, asmc_stream stream);

/* w_i = exp((lw_i(beta) + shift) - lse)  — the normalised weights of samples.py:1277. */
int asmc_normalized_weights(asmc_ctx* ctx, int64_t n, const double* ll_dev, const double* lp_dev,
                            const double* lq_dev, double beta0, double beta, double shift,
                            double lse, double* w_out_dev, asmc_stream stream);

/* NaN / non-finite guards (smc/base.py:330-335, mcmc.py:68-74, minipcn.py:133-134). */
int asmc_count_nonfinite(asmc_ctx* ctx, int64_t n, const double* v_dev, int64_t* n_nan_host,
                         int64_t* n_inf_host, asmc_stream stream);

/* ---- resampling (reference samples.py:1277-1287; numpy Generator.choice semantics) ---------
 * asmc_cdf: inclusive cumulative sum of w, carry_in added in front (rank chaining).
 *   mode ASMC_CDF_EXACT reproduces the *sequential* fp64 accumulation of numpy's cumsum
 *   bit-for-bit (parallelised through round-to-nearest-even integer transducers);
 *   ASMC_CDF_FAST uses a parallel scan order.  total_host receives the last element.
 * asmc_cdf_normalize: cdf /= last    (numpy: cdf /= cdf[-1]).
 * asmc_cdf_normalize_last: the same with the total the preceding asmc_cdf on this ctx left on the device
 *   (call asmc_cdf with total_host = NULL: no host round trip between the scan and the division).
 * asmc_pcg64_uniforms: u[j] = j-th next double of numpy's PCG64 stream given its raw state
 *   {state_hi, state_lo, inc_hi, inc_lo} after skipping `offset` draws (the host then calls
 *   bit_generator.advance(n)).
 * asmc_search: idx[j] = #{k : cdf[k] <= u[j]}  == cdf.searchsorted(u, side="right").
 * asmc_gather: x_out[j,:] = x_in[idx[j],:] and the three scalar vectors (samples.py:1279-1287). */
int asmc_cdf(asmc_ctx* ctx, int64_t n, const double* w_dev, double* cdf_dev, int mode,
             double carry_in, double* total_host, asmc_stream stream);
int asmc_cdf_normalize(asmc_ctx* ctx, int64_t n, double* cdf_dev, double last, asmc_stream stream);
int asmc_cdf_normalize_last(asmc_ctx* ctx, int64_t n, double* cdf_dev, asmc_stream stream);
int asmc_cdf_total_dev(asmc_ctx* ctx, double* out_dev, asmc_stream stream); /* total of the last asmc_cdf, device to device */
int asmc_pcg64_uniforms(asmc_ctx* ctx, const uint64_t state_host[4], uint64_t offset, int64_t n,
                        double* u_dev, asmc_stream stream);
/* Sharded exact cdf (one process per GPU; the reference has no distributed mode): this rank's slice of
 * numpy's SEQUENTIAL cumsum over the GLOBAL weight vector (ranks in order), divided by the global total - the values
 * Generator.choice searches (samples.py:1277-1278) - without a rank waiting for another rank's scan:
 *   asmc_cdf_shard_records  passes that need only an APPROXIMATE incoming sum (approx_carry = sum of the lower ranks'
 *                           weights to a few ulps; ignored on the first rank, whose first tile is scanned exactly from 0
 *                           into cdf_dev): one ASMC_CDF_REC-word record per 2048-particle tile -> rec_dev
 *                           [asmc_cdf_shard_tiles(n) * ASMC_CDF_REC];
 *   (caller)                all-gather of the records, concatenated in rank order -> recs_all_dev;
 *   asmc_cdf_shard_chain    every rank walks the same chain over ALL tiles, verifying each record against the exact
 *                           running sum; tile0 = global index of this rank's first tile; work_dev = 3 * n_tiles_total
 *                           doubles of scratch kept between the calls.  A tile whose record fails verification (always
 *                           possible where the sum approaches a power of two, i.e. at the very end of normalised weights)
 *                           is scanned element-wise by the rank that owns it, which publishes the exact sum behind it in
 *                           state_out_dev[ASMC_CDF_STATE]; other ranks stop there.  Round 1: states_all_dev = NULL.
 *   (caller)                all-gather of the states -> states_all_dev [world * ASMC_CDF_STATE];
 *   asmc_cdf_shard_chain    round 2 (states_all_dev given): resumes and passes foreign tiles through the published sums;
 *   asmc_cdf_shard_finish   writes this rank's slice divided by the global total from its final state (state_dev =
 *                           round 2's state_out_dev); out_dev[4] = {fail, global total, lo, hi}: [lo, hi) is this
 *                           shard's slice of the normalised cdf (lo = cdf of the last particle of the rank below, 0 on the
 *                           first rank; hi = cdf_dev[n - 1]).
 * fail != 0: the chain could not be completed in two rounds (failing tiles on several ranks blocking each other): cdf_dev
 * is undefined and the caller recomputes with a replicated asmc_cdf over the all-gathered weights.  None of the calls
 * synchronises.
 * asmc_select_range: out = u[(lo <= u) & (u < hi)] in index order, lohi_dev = {lo, hi} on the device (out_dev + 2 of the
 *   call above); out_dev must hold n doubles; count_host receives the number kept (synchronises).  With u = the draws of
 *   asmc_pcg64_uniforms this is the sub-sequence of Generator.choice's draws whose ancestors live on this rank. */
int64_t asmc_cdf_shard_tiles(int64_t n);
int asmc_cdf_shard_records(asmc_ctx* ctx, int64_t n, const double* w_dev, double* cdf_dev, double approx_carry,
                           int first_rank, int64_t* rec_dev, asmc_stream stream);
int asmc_cdf_shard_records_dev(asmc_ctx* ctx, int64_t n, const double* w_dev, double* cdf_dev,
                               const double* approx_carry_dev, const double* tile_sums_dev, int first_rank, int64_t* rec_dev,
                               asmc_stream stream);
int asmc_cdf_shard_chain(asmc_ctx* ctx, int64_t n, const double* w_dev, double* cdf_dev, const int64_t* recs_all_dev,
                         int64_t n_tiles_total, int64_t tile0, double* work_dev, const double* states_all_dev, int world,
                         int rank, double* state_out_dev, asmc_stream stream);
int asmc_cdf_shard_finish(asmc_ctx* ctx, int64_t n, const double* w_dev, double* cdf_dev, const int64_t* recs_all_dev,
                          int64_t n_tiles_total, int64_t tile0, double* work_dev, const double* state_dev, double* out_dev,
                          asmc_stream stream);
int asmc_cdf_shard_finish_select(asmc_ctx* ctx, int64_t n, const double* w_dev, double* cdf_dev, const int64_t* recs_all_dev,
                                 int64_t n_tiles_total, int64_t tile0, double* work_dev, const double* state_dev, int64_t n_u,
                                 const double* u_dev, double* edges_out_dev, double* out_dev, int64_t* info_dev,
                                 asmc_stream stream);
int asmc_select_range(asmc_ctx* ctx, int64_t n, const double* u_dev, const double* lohi_dev, double* out_dev,
                      int64_t* count_host, asmc_stream stream);
/* the same selection, enqueued only: info_dev[0] = kept count, info_dev[1] = the failure flag edges_dev[0] of the slice
 * (edges_dev = {fail, total, lo, hi} as asmc_cdf_shard_finish leaves them) - the caller all-gathers info_dev over the ranks
 * and reads all of it back with one synchronisation */
int asmc_select_range_dev(asmc_ctx* ctx, int64_t n, const double* u_dev, const double* edges_dev, double* out_dev,
                          int64_t* info_dev, asmc_stream stream);

/* Sharded multinomial resampling with the offspring kept on the ancestor's rank: every rank walks
 * the SAME n_total draws of the PCG64 stream and keeps those inside its own slice [lo, hi) of the global cdf, mapped
 * to its local cdf's coordinate q = (u - lo) / (hi - lo) in [0, 1).
 *   asmc_pcg64_select          generate + filter into stage_dev[asmc_pcg64_select_stage_len(n_total)]; count_host
 *                              receives the number kept (synchronises);
 *   asmc_pcg64_select_compact  the kept q's, dense, into q_dev[count].
 * Order of the kept draws (part of the specification; tests/oracle restate it): draw i is generated by thread
 * j = i % ASMC_SELECT_THREADS in iteration k = i / ASMC_SELECT_THREADS; kept draws are ordered by (j / 64, k, j % 64). */
int64_t asmc_pcg64_select_stage_len(int64_t n_total);
int asmc_pcg64_select(asmc_ctx* ctx, const uint64_t state_host[4], int64_t n_total, double lo, double hi,
                      double* stage_dev, int64_t* count_host, asmc_stream stream);
int asmc_pcg64_select_compact(asmc_ctx* ctx, int64_t n_total, const double* stage_dev, double* q_dev,
                              asmc_stream stream);
int asmc_systematic_uniforms(asmc_ctx* ctx, int64_t n_out, int64_t j0, int64_t n_total, double u0,
                             const double* v_dev /* NULL: systematic; else stratified */,
                             double* u_dev, asmc_stream stream);
int asmc_search(asmc_ctx* ctx, int64_t n, const double* cdf_dev, int64_t n_out, const double* u_dev,
                int64_t* idx_dev, asmc_stream stream);
/* x_out[j] = x_in[idx[j]] and the same for the three log-probability vectors (samples.py:1278-1287, `self[idx]`).
 * n_in = rows of the source population (idx values are < n_in): when the population fits the ctx and most of it is
 * drawn, (ll, lp, lq) are first packed into one 32-byte record per particle, so that a draw costs ONE random sector
 * read for its three scalars instead of three.  asmc_importance_step packs those records on its way: an asmc_gather of
 * the SAME three arrays and n_in that follows it with no other call on this ctx in between skips the packing pass - the
 * caller must not have rewritten the arrays between the two calls (any library launch in between drops the records). */
int asmc_gather(asmc_ctx* ctx, int64_t n_in, int64_t n_out, const int64_t* idx_dev, int d, int x_dtype,
                const void* x_in_dev, void* x_out_dev, const double* ll_in_dev,
                const double* lp_in_dev, const double* lq_in_dev, double* ll_out_dev,
                double* lp_out_dev, double* lq_out_dev, asmc_stream stream);

/* ---- proposal draw / built-in densities / validity filter (reference mcmc.py:49-110) --------
 * asmc_gaussian_draw: analytic diagonal-Gaussian proposal (a `Flow`-interface object with
 *   sample_and_log_prob, flows/base.py:11-98): x = mu + sigma * xi (Philox4x32-10 normals, keyed
 *   by seed; counter = global particle index gid0+i, draw id), lq = log N(x; mu, diag sigma^2).
 * asmc_mixture_logpdf: evaluate a built-in density over the batch.
 * asmc_compact_valid: keep rows with finite lp and ll, preserving order (mcmc.py:88-90);
 *   n_valid_host receives the count; outputs must hold n rows. */
int asmc_gaussian_draw(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const double* mu_dev,
                       const double* sigma_dev, uint64_t seed, uint64_t gid0, uint32_t draw_id,
                       void* x_out_dev, double* lq_out_dev, asmc_stream stream);
int asmc_mixture_logpdf(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev,
                        const asmc_mixture* density, double* out_dev, asmc_stream stream);
/* The same density evaluated behind a per-coordinate affine map and clamp, plus a quadratic term:
 *   out_i = log sum_c w_c N(t_i; mu_c, diag prec_c^-1) + sum_j h_j t_ij^2,   t_ij = clip(a_j x_ij + b_j, lo_j, hi_j),
 * premap_dev = rows a, b, lo, hi, h of d doubles.  It is the proposal flow's log q(x') written as a function of the
 * preconditioned coordinate z' when the flow's data transform and the preconditioning transform share their bounded ->
 * unbounded stage (reference flows/torch/flows.py:368-387 behind transforms.py:270-316): the probit / erfinv round trip
 * z' -> x' -> flow latent collapses to an affine map, the clamp reproducing the forward transform's eps clip.
 * Rows must be a power-of-two number (<= 64) of 16-byte pieces, <= 4 components. */
int asmc_mixture_logpdf_premap(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev, const double* premap_dev,
                               const asmc_mixture* density, double* out_dev, asmc_stream stream);
/* asmc_compact_valid: *n_valid_host == n means the population was already compact and NOTHING was copied - the caller keeps
 * using its inputs (the copy of a 1M x 32 population is 0.6 ms; draw_initial_samples almost always lands here). */
int asmc_compact_valid(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev,
                       const double* ll_dev, const double* lp_dev, const double* lq_dev,
                       void* x_out_dev, double* ll_out_dev, double* lp_out_dev, double* lq_out_dev,
                       int64_t* n_valid_host, asmc_stream stream);

/* ---- pCN mutation (replaces minipcn via smc/minipcn.py:69-135; spec in DESIGN.md) -----------
 * asmc_colsum / asmc_centered_gram: population moments for the reference Gaussian
 *   (sum_i x_i ; sum_i (x_i - c)(x_i - c)^T, full d x d, row-major) — per-rank partial sums.
 * asmc_pcn_mutate: n_steps fused pCN steps (propose + built-in target + accept) in place.
 *   rho_inout_host: step size in/out; n_accept_host[n_steps] accepted counts per step
 *   (LOCAL particles); rho_hist_host[n_steps] step size used at each step (may be NULL).
 *   With params->adapt the step size is adapted on device after every step, from the LOCAL acceptance rate
 *   or - sharded runs - from the GLOBAL one once asmc_pcn_set_count_hook has installed an exchange hook.
 * asmc_pcn_set_count_hook: sharded mutation without host round trips.  After each step's kernel the library sums
 *   this rank's accept count into cell_dev[0] (int64) and calls hook(user, stream); the hook must enqueue an in-place
 *   SUM of cell_dev over all ranks on `stream` (RCCL all-reduce) and return 0.  The adaptation kernel then divides by
 *   n_global, and n_accept_host reports GLOBAL counts.  hook == NULL removes it.  Applies to asmc_pcn_mutate and
 *   asmc_pcn_mutate_flow.
 * asmc_set_rccl: hands the library a communicator of its own for the collectives it issues ITSELF, on the stream its
 *   kernels run on (no stream hop, no host callback).  The library does not link RCCL: the caller passes the address of
 *   the process's own ncclAllReduce and an ncclComm_t it has created (one rank per GPU); NULL, NULL removes them.
 * asmc_pcn_set_count_rccl: the accept-count exchange as ncclAllReduce(cell, cell, 1, ncclInt64, ncclSum) on that
 *   communicator: a step boundary is one small RCCL kernel.  cell_dev == NULL removes it.
 * asmc_pcn_propose / asmc_pcn_accept: the split form for arbitrary Python callables / torch flows
 *   (the host evaluates log_q, log_prior, log_likelihood on x_prop between the two calls,
 *   reference smc/base.py:507-519).  logj_old_dev / logj_new_dev (both or neither): log|det J| of the
 *   preconditioning transform at the current / proposed state, added to the tempered log-target
 *   (smc/base.py:515-517); logj_old_dev is updated in place for accepted particles.
 * Non-finite log-targets (every step entry point; ABI 23): the tempered log-target (1 - beta) log q + beta (ll + lp) [+ log|det J|]
 *   is mapped to -inf when it is NaN - the reference's rule, smc/base.py:518 - AND when it is +inf: such a proposal is rejected and
 *   counted as a rejection, and no particle settles on log_likelihood = +inf (the next log-sum-exp would be NaN and the temperature
 *   search would stall).  That is this library's reading of what the third-party step the reference calls (minipcn, absent here) must
 *   do for the reference's own +inf likelihood-hole test to finish (tests/integration_tests/test_integration.py:131-166);
 *   unverified against the package.  A carried +inf (caller-supplied state) loses against any finite proposal. */
/* Student-t reference of the tpCN step (params->nu > 0): the EM fit of (mu, Sigma, nu) runs on a subsample of m particles
 * (xs_dev [m, d] fp64, gathered by the caller); these calls are its per-particle half, the host keeps the d x d algebra.
 * asmc_student_estep: y = Linv (x - mu), z = (nu + d)/(nu + |y|^2) -> z_dev[m];
 *   sums_host[d + 2] = { sum z, sum (log z - z), sum z x[0..d) } (block partials added in block order).
 * asmc_student_scale: r = sqrt(z) (x - mu) -> r_dev[m, d]; asmc_centered_gram(r, center 0) is then the weighted scatter. */
int asmc_student_estep(asmc_ctx* ctx, int64_t m, int d, const double* xs_dev, const double* mu_host,
                       const double* linv_host /* [d, d] row-major, lower triangle */, double nu, double* z_dev,
                       double* sums_host, asmc_stream stream);
int asmc_student_scale(asmc_ctx* ctx, int64_t m, int d, const double* xs_dev, const double* z_dev, const double* mu_host,
                       double* r_dev, asmc_stream stream);
/* The whole EM of the Student-t reference on the stream (student_t.py fit_student_t_device: initial moments, then per iteration
 * the factorisation of the scale matrix, E-step, weighted mean, degrees of freedom, scatter matrix), one synchronisation at the
 * end instead of three per iteration; iterations behind the one that meets |nu' - nu| <= rtol nu return at once.  d in {32, 64,
 * 128}, rows 16-byte aligned, m <= ASMC_STUDENT_MAX_ROWS.  r_scratch_dev [m, d], z_scratch_dev [m]: scratch.  out_dev: (mu | L |
 * Linv) of the final fit in asmc_reference_factor's layout.  result_host [4 + d + d * d]: nu (clamped to [1, 1e6]), iterations,
 * factorisation status (0 / -1), converged, mu, Sigma. */
int asmc_student_fit(asmc_ctx* ctx, int64_t m, int d, const double* xs_dev, int max_iter, double rtol, double nu0,
                     double* r_scratch_dev, double* z_scratch_dev, double* out_dev, double* result_host, asmc_stream stream);
typedef int (*asmc_count_hook)(void* user, asmc_stream stream);
int asmc_pcn_set_count_hook(asmc_ctx* ctx, asmc_count_hook hook, void* user, int64_t* cell_dev, int64_t n_global);
/* Lagged step-size adaptation (asmc_pcn_params.adapt = k >= 2; sampler_kwargs["adapt_lag"]): the step size is held for blocks
 * of k steps and the block's k updates are applied at its end, in order, each with its own step's count - the arithmetic of
 * adapt = 1, k steps late.  A sharded run then exchanges accept counts once per block: the hook must sum n_cells >= k
 * consecutive int64 cells starting at cell_dev (step t of a block writes cell t % k), which this call announces
 * (<= ASMC_MAX_COUNT_CELLS; asmc_pcn_set_count_hook resets it to 1).  Honoured by asmc_pcn_mutate and asmc_pcn_mutate_flow;
 * the one-step-at-a-time entry points (asmc_pcn_split_*, asmc_pcn_ysplit_*) adapt after every step. */
#define ASMC_MAX_COUNT_CELLS 64
int asmc_pcn_set_count_cells(asmc_ctx* ctx, int n_cells);
/* NaNs in the carried log q after the last asmc_pcn_mutate / asmc_pcn_mutate_flow call (the reference's check after every
 * mutation, smc/minipcn.py; counted on the device and read back with the call's own results: no extra synchronisation) */
int64_t asmc_pcn_lq_nan(asmc_ctx* ctx);
int asmc_set_rccl(asmc_ctx* ctx, void* allreduce_fn, void* nccl_comm);
int asmc_set_rccl_allgather(asmc_ctx* ctx, void* allgather_fn); /* the process's ncclAllGather, for asmc_find_beta_shard_rounds */
/* equal-sized all-gather of `count` 8-byte elements per rank (fp64, or int64 with is_int64) on that communicator and stream */
int asmc_rccl_all_gather(asmc_ctx* ctx, const void* send_dev, void* recv_dev, int64_t count, int is_int64, asmc_stream stream);
int asmc_pcn_set_count_rccl(asmc_ctx* ctx, int64_t* cell_dev, int64_t n_global);
int asmc_colsum(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev, double* sum_host,
                asmc_stream stream);
/* asmc_colsum followed by asmc_centered_gram around sum / n_mean, as one enqueue with one synchronisation (the centre stays
 * on the device; the results equal the two calls' bit for bit).  across_ranks != 0 (sharded populations, after asmc_set_rccl):
 * both results are summed over the ranks by ncclAllReduce on the stream - every rank receives the same bits - and n_mean is
 * the global population; fp64-MFMA shapes only (d in {32, 64, 128} rows 16-byte aligned), other shapes return an error and
 * the caller merges the two calls' partials on the host.  Reference: the population moments behind the pCN reference
 * Gaussian, smc/minipcn.py:75-84. */
int asmc_mean_gram(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev, int64_t n_mean, int across_ranks,
                   double* sum_host, double* gram_host /* [d, d] */, asmc_stream stream);
/* The two halves of asmc_mean_gram for a caller that has more on the stream: _enqueue starts both passes and their copies to
 * pinned memory (fp64-MFMA shapes only, error otherwise), _fetch synchronises the stream and hands the results out.  One
 * pending request per context. */
/* across_ranks is a set of flags: ASMC_GRAM_ACROSS_RANKS (1) as above; ASMC_GRAM_FROM_GATHER (2): x_dev are the rows the last
 * asmc_gather on this ctx wrote and NOTHING has rewritten them since (the caller vouches for its own kernels; any library launch
 * in between is noticed) - the column sums that rode along the gather are used and the pass over the rows is skipped. */
#define ASMC_GRAM_ACROSS_RANKS 1
#define ASMC_GRAM_FROM_GATHER 2
int asmc_mean_gram_enqueue(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev, int64_t n_mean, int across_ranks,
                           asmc_stream stream);
int asmc_mean_gram_fetch(asmc_ctx* ctx, int d, double* sum_host, double* gram_host /* [d, d] */, asmc_stream stream);
/* The reference Gaussian of the coming mutation from those moments WITHOUT a host round trip (smc/minipcn.py:75-84; what the
 * host did with numpy: cov = G / (n_cov - 1), symmetrised, L = chol(cov + jitter mean(diag) I) with the jitter ladder 0, 1e-12,
 * 1e-10, ... of twelve tries, Linv = L^-1): one block on the stream behind the Gram kernel.  sum_host = gram_host = NULL: the
 * moments of the pending asmc_mean_gram_enqueue (consumed - no _fetch follows); otherwise moments the caller merged on the
 * host, uploaded first.  out_dev: [mu | pad to seg | L (d x d row-major, zeros above the diagonal) | pad to seg x d | Linv],
 * seg = 32 ceil(d / 32) doubles.  d <= 128.  asmc_reference_factor_status (after the caller's next synchronisation of the
 * stream): jitter tries used (0: none), -1: not factorable / not finite, -2: not synchronised yet. */
int asmc_reference_factor(asmc_ctx* ctx, int d, int64_t n_mean, int64_t n_cov, const double* sum_host, const double* gram_host,
                          double* out_dev, asmc_stream stream);
int asmc_reference_factor_status(asmc_ctx* ctx, int* status_host);
/* the generation number of the request just made (1, 2, ...) and the status of that particular request: a caller that keeps
 * the next temperature's factorisation on the stream behind a mutation asks about the one that served the mutation (valid for
 * the last 15 requests) */
int64_t asmc_reference_factor_generation(asmc_ctx* ctx);
int asmc_reference_factor_status_of(asmc_ctx* ctx, int64_t generation, int* status_host);
/* The same fit for a sharded population without a host round trip: the rank's column sums / centred Gram matrix (centre =
 * sum_dev / n_mean, i.e. the GLOBAL sums and population once the caller has all-reduced sum_dev) land in the caller's device
 * buffers, the caller sums them over the ranks on the stream (torch.distributed all_reduce = RCCL), and asmc_reference_factor_dev
 * factors the result.  asmc_centered_gram_dev: d in {32, 64, 128}, 16-byte aligned rows (ASMC_ERR_UNSUPPORTED otherwise). */
int asmc_colsum_dev(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev, int from_gather /* see ASMC_GRAM_FROM_GATHER */,
                    double* sum_dev /* [d] */, asmc_stream stream);
int asmc_centered_gram_dev(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev, const double* sum_dev, int64_t n_mean,
                           double* gram_dev /* [d, d] */, asmc_stream stream);
int asmc_reference_factor_dev(asmc_ctx* ctx, int d, int64_t n_mean, int64_t n_cov, const double* sum_dev, const double* gram_dev,
                              double* out_dev, asmc_stream stream);
int asmc_centered_gram(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev,
                       const double* center_host, double* gram_host, asmc_stream stream);
int asmc_pcn_mutate(asmc_ctx* ctx, int64_t n, void* x_dev, double* ll_dev, double* lp_dev,
                    double* lq_dev, const asmc_pcn_params* params, int n_steps, uint32_t step0,
                    double* rho_inout_host, int64_t* n_accept_host, double* rho_hist_host,
                    asmc_stream stream);
int asmc_pcn_propose(asmc_ctx* ctx, int64_t n, int d, int x_dtype, const void* x_dev,
                     void* x_prop_dev, double* qform_old_dev, double* qform_new_dev,
                     const double* mu_dev, const double* L_dev, const double* Linv_dev, double rho,
                     double nu /* as asmc_pcn_params.nu; the qform outputs are then (d + nu) log(1 + |y|^2/nu) */,
                     uint64_t seed, uint64_t gid0, uint32_t step, asmc_stream stream);
/* The split form without a host round trip per step (the host may enqueue step t + 1 while step t runs):
 *   asmc_pcn_split_begin(rho0)                 step size into the ctx;
 *   asmc_pcn_propose(..., rho = 0, ...)        uses the device-resident step size;
 *   asmc_pcn_accept(..., n_accept_host = NULL) leaves the accept count on the device;
 *   asmc_pcn_split_adapt(n_global, target, t, adapt)   closes step t: count (summed over ranks through the exchange hook
 *                                              when one is installed) -> history, step-size adaptation, all on the stream;
 *   asmc_pcn_split_end(n_steps, ...)           counts, step-size history and final step size to the host (synchronises). */
int asmc_pcn_split_begin(asmc_ctx* ctx, double rho0, asmc_stream stream);
int asmc_pcn_split_adapt(asmc_ctx* ctx, int64_t n_global, double target_accept, int t, int adapt, asmc_stream stream);
int asmc_pcn_split_end(asmc_ctx* ctx, int n_steps, int64_t* n_accept_host, double* rho_hist_host, double* rho_host,
                       asmc_stream stream);
/* Whitened-state session of the split form (d in {4, 8, 16, 32}; otherwise ASMC_ERR_UNSUPPORTED and the caller uses
 * asmc_pcn_propose / asmc_pcn_accept): the chain state is kept as y = L^-1 (x - mu), coordinate-major, in the context for
 * the length of the mutation, so a step needs one mat-vec and no LDS staging of the state.
 *   asmc_pcn_ysplit_begin    x -> y (x_dev is not modified), step size rho0 into the ctx;
 *   asmc_pcn_ysplit_propose  y' = sqrt(1 - rho^2) y + rho sqrt(s) xi (not stored), x' = mu + L y' -> x_prop_dev [n, d];
 *   (caller)                 log_q, log_prior, log_likelihood at x_prop (Python callables, flows, ...);
 *   asmc_pcn_ysplit_accept   regenerates y' from the same counters, accepts (y <- y', carried log-probabilities updated) and
 *                            closes step t like asmc_pcn_split_adapt (count exchange hook, history, adaptation);
 *                            logj_dev / logj_new_dev (both or NULL): carried / proposed log|det J| of a chain that runs in a
 *                            preconditioned space z = T(x) - they join the tempered log-target and follow the accepted
 *                            state, as in asmc_pcn_accept (smc/base.py:507-519);
 *   asmc_pcn_ysplit_end      y -> x_dev;  asmc_pcn_split_end then returns counts, step-size history and final step size.
 * params: d, x_dtype, beta, mu / L / Linv, seed, gid0, target_accept, adapt, nu, noise are used (the same values in every
 * call of a session); the mixtures are ignored.  A context carries one mutation at a time (step size, accept counts,
 * scale variates and the whitened state live in it): the caller's densities may use every other entry point on the same
 * context between the calls (mixture / coupling log-densities, transforms, reductions) but not another asmc_pcn_* call;
 * all calls of a session go to the same stream. */
int asmc_pcn_ysplit_begin(asmc_ctx* ctx, int64_t n, const void* x_dev, const asmc_pcn_params* params, double rho0,
                          asmc_stream stream);
int asmc_pcn_ysplit_propose(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* params, uint32_t step, void* x_prop_dev,
                            asmc_stream stream);
int asmc_pcn_ysplit_accept(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* params, uint32_t step, double* ll_dev,
                           double* lp_dev, double* lq_dev, const double* ll_new_dev, const double* lp_new_dev,
                           const double* lq_new_dev, double* logj_dev, const double* logj_new_dev, int64_t n_global, int t,
                           asmc_stream stream);
int asmc_pcn_ysplit_end(asmc_ctx* ctx, int64_t n, void* x_dev, const asmc_pcn_params* params, asmc_stream stream);
int asmc_pcn_accept(asmc_ctx* ctx, int64_t n, int d, int x_dtype, void* x_dev,
                    const void* x_prop_dev, double* ll_dev, double* lp_dev, double* lq_dev,
                    const double* ll_new_dev, const double* lp_new_dev, const double* lq_new_dev,
                    double* logj_old_dev, const double* logj_new_dev,
                    const double* qform_old_dev, const double* qform_new_dev, double beta,
                    uint64_t seed, uint64_t gid0, uint32_t step, int64_t* n_accept_host,
                    asmc_stream stream);

/* ---- coupling-flow log-density ------------------------------------------
 * Replaces the torch round trip of Flow.log_prob (flows/torch/flows.py:368-387) inside the tempered
 * log-target of each MCMC step (smc/base.py:507-519) and in draw_initial_samples (mcmc.py:49-110).
 * The flow is an affine coupling flow (RealNVP): x' = (x - loc) / scale, then n_layers coupling layers
 * whose masks alternate first half / second half; each conditioner is the MLP
 * d/2 -> hidden -> hidden -> d (ReLU), its output split into (s_raw, t) for the d/2 transformed
 * coordinates; s = 2 tanh(s_raw / 2), z_b = (x_b - t) exp(-s);
 * log q(x) = -|z|^2 / 2 - d/2 log(2 pi) - sum s - sum log(scale).  fp32 arithmetic on the fp32-input
 * MFMA, result widened to fp64.  zuko (the reference's flow library) is absent: parity unpinned.
 *   asmc_coupling_pack_floats: length of the packed parameter block, or <0 when the shape is
 *     unsupported (dims even, <= 64; hidden in {32, 64, 128}).
 *   asmc_coupling_pack (host only, no GPU): weights_host[3c+0..2] / biases_host[3c+0..2] are the
 *     three dense layers of coupling layer c in torch.nn.Linear layout ([out, in] row-major;
 *     [hidden, d/2], [hidden, hidden], [d, hidden] with output rows s_raw_0.. then t_0..);
 *     writes them in MFMA operand order into packed_host.
 *   asmc_coupling_logprob: out_dev[i] = log q(x_i). */
#define ASMC_FLOW_COUPLING 0 /* affine coupling layers with alternating half masks (above) */
#define ASMC_FLOW_MAF 1      /* masked autoregressive transforms (below) */
typedef struct asmc_coupling {
    int32_t dims;
    int32_t n_layers; /* coupling layers, or autoregressive transforms */
    int32_t hidden;
    int32_t kind;     /* ASMC_FLOW_COUPLING / ASMC_FLOW_MAF: every entry point that takes a flow dispatches on it */
    const float* packed_dev; /* asmc_coupling_pack output copied to the device */
    const float* loc_dev;    /* [dims] */
    const float* scale_dev;  /* [dims] */
    double log_scale_sum;    /* sum_j log(scale_j) */
    int32_t affine;          /* how (s_raw, t) act on a coordinate: 0 = this repository's flows, s = 2 tanh(s_raw / 2),
                                z = (x - t) exp(-s);  ASMC_AFFINE_SOFTCLIP = zuko's MonotonicAffineTransform (slope 1e-3):
                                ls = s_raw / (1 + |s_raw| / ln 1000), z = x exp(ls) + t - for flows trained by the reference
                                (flows/torch/flows.py:156-168; aspire_amd.flows.MAFFlow.from_zuko_state_dict).  zuko is absent
                                from the build image: that form follows its documented arithmetic and is UNVERIFIED against it.
                                Autoregressive flows only. */
    int32_t reserved;
} asmc_coupling;
#define ASMC_AFFINE_TANH 0
#define ASMC_AFFINE_SOFTCLIP 1
/* Packed layout of a flow of this shape: 0 = 32-particle tiles with every layer resident in LDS (dims <= 32), 1 = 16-particle
 * groups with the weights streamed through LDS (32 < dims <= 128: csrc/asmc_flow16.hip; coupling flows need even dims; and -
 * ABI 23 - autoregressive flows of hidden width 128 at ANY dims <= 128: the resident layout holds one such transform at most),
 * < 0 = no kernel.  The pack functions below choose it from the shape; every entry point that takes an asmc_coupling follows.
 * The one-kernel mutation step (asmc_pcn_mutate_flow) exists for hidden widths 32 / 64 / 128 in layout 1 (ABI 23; width 64 only
 * before) and for 32 / 64 - coupling flows: 128 too, while the layers fit the LDS - in layout 0. */
int asmc_flow_layout(int kind, int dims, int hidden);
int64_t asmc_coupling_pack_floats(int dims, int n_layers, int hidden);
int asmc_coupling_pack(int dims, int n_layers, int hidden, const float* const* weights_host,
                       const float* const* biases_host, float* packed_host);
int asmc_coupling_logprob(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x_dev,
                          const asmc_coupling* flow, double* out_dev, asmc_stream stream);
/* The sampling direction of the same flow (reference flows/torch/flows.py:327-346, Flow.sample_and_log_prob; the proposal draw
 * of mcmc.py:49-110): n latent draws z ~ N(0, I) from the counter-based generator (fp32 Box-Muller; key `seed`, counter = global
 * particle index gid0 + i and draw_id, so a draw does not depend on how the population is sharded), pushed through the inverted
 * coupling layers on the matrix cores; x_out_dev [n, dims] in x_dtype and log q(x) of every sample from the same pass.
 * Split-fp16 layers with every layer resident in LDS; ASMC_ERR_UNSUPPORTED otherwise (the caller samples with its own modules). */
int asmc_coupling_sample(asmc_ctx* ctx, int64_t n, int x_dtype, const asmc_coupling* flow, uint64_t seed, uint64_t gid0,
                         uint32_t draw_id, void* x_out_dev, double* lq_out_dev, asmc_stream stream);

/* ---- masked autoregressive flow (MAF), the reference's DEFAULT flow class ---------------------------------------------
 * ZukoFlow(flow_class="MAF") (flows/torch/flows.py:140-168) evaluated by log_prob (:368-387) in every MCMC step and by
 * sample_and_log_prob (:327-346) in the proposal draw.  zuko is absent: the architecture is this repository's statement of
 * Papamakarios et al. 2017 (aspire_amd/flows.py MAFFlow) - x' = (x - loc) / scale, then n_layers autoregressive transforms
 * z_i = (x_i - t_i(x_<i)) exp(-s_i(x_<i)), s = 2 tanh(s_raw / 2), with (s_raw, t) = MLP dims -> hidden -> hidden -> 2 dims
 * (ReLU) whose weights carry the MADE masks (the masks, and with them each transform's variable order, are folded into the
 * weights the caller hands over: the kernels see dense layers);  log q(x) = N(z; 0, I) - sum s - sum log scale.
 * A transform is evaluated exactly like a coupling layer whose conditioner input AND transformed block are the whole x: the
 * same split-fp16 MFMA layers, the same packing with the roles "dims / 2 -> dims".  dims <= 32 (odd dims allowed), hidden in
 * {32, 64, 128}; the density is ONE pass per transform, sampling inverts a transform by `dims` passes (each pass fixes the
 * coordinates whose inputs are already final; intermediate values are clamped to the fp16 operand range so that a
 * not-yet-final coordinate cannot poison the others through a masked - zero - weight), used once per run.
 *   asmc_maf_pack_floats / asmc_maf_pack (host only): weights_host[3t+0..2] / biases_host[3t+0..2] = the three MASKED dense
 *     layers of transform t in torch.nn.Linear layout ([hidden, dims], [hidden, hidden], [2 dims, hidden] with output rows
 *     s_raw_0 .. s_raw_{dims-1}, t_0 .. t_{dims-1}).
 * The packed block goes into an asmc_coupling with kind = ASMC_FLOW_MAF; asmc_coupling_logprob, asmc_coupling_sample,
 * asmc_pcn_mutate_flow(_enqueue) take it as they take a coupling flow (the fused one-kernel step at every dims <= 32). */
int64_t asmc_maf_pack_floats(int dims, int n_transforms, int hidden);
int asmc_maf_pack(int dims, int n_transforms, int hidden, const float* const* weights_host, const float* const* biases_host,
                  float* packed_host);

/* asmc_pcn_mutate_flow: asmc_pcn_mutate with the proposal density q given by a coupling flow instead
 * of a built-in mixture (params->log_q is ignored): per step propose -> log q(x') on the MFMA ->
 * built-in log prior / log likelihood -> accept -> device-side adaptation, all enqueued without host
 * round trips (smc/minipcn.py:97-114 + smc/base.py:507-519 with a flow proposal, BASELINE config 3).
 * work_dev: caller-owned scratch of asmc_pcn_flow_work_bytes(n, d, x_dtype) bytes. */
int64_t asmc_pcn_flow_work_bytes(int64_t n, int d, int x_dtype);
int asmc_pcn_mutate_flow(asmc_ctx* ctx, int64_t n, void* x_dev, double* ll_dev, double* lp_dev,
                         double* lq_dev, const asmc_pcn_params* params, const asmc_coupling* flow,
                         void* work_dev, int64_t work_bytes, int n_steps, uint32_t step0,
                         double* rho_inout_host, int64_t* n_accept_host, double* rho_hist_host,
                         asmc_stream stream);
/* The same call in two halves, for a caller that has more to put on the stream behind the mutation (the next temperature's
 * importance step, asmc_importance_step): _enqueue starts everything including the read-back into pinned memory and returns,
 * _result waits for the mutation's own read-back (an event recorded behind it - not for what the caller has enqueued since) and
 * hands the results out.  One pending mutation per context; the tables the kernels read
 * (params, flow) must stay alive until _result.
 * Ordering of x_dev: the blocking form returns with x_dev, ll / lp / lq final (it synchronises the stream behind every kernel
 * that writes them, the un-padding copy of a d < 32 problem included).  After _result the scalars are final and x_dev / ll /
 * lp / lq are STREAM-ORDERED: work enqueued on `stream` sees them, another stream or the host must synchronise `stream`. */
int asmc_pcn_mutate_flow_enqueue(asmc_ctx* ctx, int64_t n, void* x_dev, double* ll_dev, double* lp_dev, double* lq_dev,
                                 const asmc_pcn_params* params, const asmc_coupling* flow, void* work_dev,
                                 int64_t work_bytes, int n_steps, uint32_t step0, double rho, asmc_stream stream);
int asmc_pcn_mutate_flow_result(asmc_ctx* ctx, int n_steps, double* rho_out_host, int64_t* n_accept_host,
                                double* rho_hist_host, asmc_stream stream);
/* Proposals of the last asmc_pcn_mutate_flow whose flow density was not finite (they were rejected).  The flow's fp32 layers
 * run as split-fp16 MFMA products (operands |.| < 65504): a badly scaled flow shows up here; ASMC_FLOW_MATH=f32 selects the
 * fp32 MFMA chain.  Counted by the one-kernel step only. */
int64_t asmc_pcn_flow_nonfinite(asmc_ctx* ctx);

/* ---- preconditioning transforms ------------------------------------------
 * Element-wise CompositeTransform of the reference (transforms.py:142-316): periodic wrap (:411-436),
 * bounded -> unbounded through logit (utils.py:196-245) or probit (transforms.py:540-571) of the unit
 * interval (:440-537), affine standardisation (:614-646); forward applies them in that order, inverse in
 * the opposite order; logj_dev[i] (may be NULL) receives log|det J| of the direction applied.
 * Per-dimension tables live in HBM: kind (0 none, 1 logit, 2 probit), periodic flag, lower, upper (finite
 * wherever kind != 0 or periodic), mean / std (both NULL: no affine stage).  The two constants are the
 * reference's scalars in FORWARD sign: unit_logj = -sum_{kind != 0} log(upper - lower),
 * affine_logj = -sum log|std| (computed by the host so that they round as the reference's do).
 * In-place operation (z_dev == x_dev) is allowed. */
#define ASMC_TR_NO_PERIODIC 1 /* hints: what the tables do NOT contain (0 = unknown); they let the kernels compile */
#define ASMC_TR_NO_LOGIT 2    /* the unused branches out (fmod, log / log1p / exp, erf / erfinv) */
#define ASMC_TR_NO_PROBIT 4
typedef struct asmc_transform {
    int32_t d;
    int32_t hints; /* ASMC_TR_NO_* bits, 0 when unknown */
    const int32_t* kind_dev;
    const int32_t* periodic_dev;
    const double* lower_dev;
    const double* upper_dev;
    const double* mean_dev;
    const double* std_dev;
    double eps;
    double unit_logj;
    double affine_logj;
} asmc_transform;
int asmc_transform_forward(asmc_ctx* ctx, int64_t n, int x_dtype, const void* x_dev, void* z_dev,
                           double* logj_dev, const asmc_transform* t, asmc_stream stream);
int asmc_transform_inverse(asmc_ctx* ctx, int64_t n, int x_dtype, const void* z_dev, void* x_dev,
                           double* logj_dev, const asmc_transform* t, asmc_stream stream);

/* asmc_pcn_ysplit_propose for a chain that lives in a preconditioned space z = T(x) (reference
 * transforms.py:294-316 inside smc/minipcn.py:105-119): the inverse transform and its log-Jacobian are applied to the
 * register-resident proposal, so the kernel emits x' = T^-1(z') (x_prop_dev) and log|det dT^-1/dz| at z' (logj_out_dev)
 * directly - no z' round trip through HBM, no separate asmc_transform_inverse pass.  `t`: a non-periodic transform with ONE
 * bounded stage (logit or probit; hints say which) and an optional affine stage.
 * premap_dev / qmix / lq_out_dev (all three or none): the proposal flow's log q(x') as a function of z' when the flow's data
 * transform shares T's bounded stage - asmc_mixture_logpdf_premap's arithmetic on a single Gaussian, minus log|J| when
 * minus_logj != 0 (the logit form; see GaussianFlow.log_prob_from_preconditioned) - evaluated in the same pass. */
int asmc_pcn_ysplit_propose_tr(asmc_ctx* ctx, int64_t n, const asmc_pcn_params* prm, uint32_t step, const asmc_transform* t,
                               const double* premap_dev, const asmc_mixture* qmix, int minus_logj, void* x_prop_dev,
                               double* logj_out_dev, double* lq_out_dev, asmc_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* ASMC_H */
