"""BASELINE.json configurations at their stated sizes on one MI355X, and the opt-in systematic / stratified resampler.

configs[2]  1M x 32, coupling-flow proposal on the fp32 MFMA + pCN mutation, 32 steps per temperature
configs[3]  the 8M-particle population of the 8-GPU configuration, on one GPU (the sharded form of the same step is
            covered by tests/test_dist_gloo.py and tests/test_gpu_dist.py)
configs[4]  1M x 128 two-component mixture target on the fp64 matrix cores, adaptive tempering
Bars: indices, gathered rows and beta* bit-exact against the CPU oracle; flow log-density at fp32 rounding;
pCN positions to 1e-9; log-evidence inside 3 sigma of the closed form (the sampler's own reported error).
"""
import math

import numpy as np
import pytest
import torch

from conftest import synth

from aspire_amd import smc_math
from aspire_amd.comm import Comm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(hip_engine):
    """The session engine grown to the 8M-particle population of configs[3]."""
    hip_engine.ensure_capacity(8_000_000, 128)
    return hip_engine


def dev(eng, *arrs):
    return tuple(eng.asarray(a) for a in arrs)


def _device_batch(eng, n, d, seed, sigma_q=1.5, dtype=torch.float64):
    """conftest.synth's recipe generated ON the device (Philox): x = sigma_q N(0, I), ll = lp = -|x|^2/2, lq = log q(x)."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.targets import DiagGaussianMixture

    flow = GaussianFlow(d, sigma=sigma_q, seed=seed, engine=eng, dtype=dtype)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    x, lq = flow.sample_and_log_prob(n)
    ll = eng.mixture_logpdf(x, lik.device_mixture(eng))
    return x, ll, ll.clone(), lq


# ---- configs[1]/[2] size: 1M x 32 exact resample, gathered ROWS against the oracle ---------------------------------
def test_exact_resample_gathers_oracle_rows_1m_d32(big, oracle):
    """samples.py:1276-1287 at BASELINE size with the real row width: the new population (x rows and the three scalar
    vectors) equals the oracle's sequential-cumsum / searchsorted / fancy-index result bit for bit."""
    from aspire_amd.samples import SMCSamples

    n, d = 1_000_000, 32
    x, ll, lp, lq = synth(n, d, 2024)
    s = SMCSamples(x=torch.as_tensor(x, device=big.device), log_likelihood=ll, log_prior=lp, log_q=lq, beta=0.0, engine=big,
                   xp=torch)
    out = s.resample(0.07, rng=np.random.default_rng(31))
    ref = oracle.resample_indices(ll, lp, lq, 0.0, 0.07, np.random.default_rng(31).random(n))
    xo, llo, lpo, lqo = oracle.gather_rows(ref, x, ll, lp, lq)
    assert np.array_equal(out.x.cpu().numpy(), xo)
    assert np.array_equal(out.log_likelihood.cpu().numpy(), llo)
    assert np.array_equal(out.log_prior.cpu().numpy(), lpo) and np.array_equal(out.log_q.cpu().numpy(), lqo)
    assert len(np.unique(ref)) < 0.8 * n  # a real resampling step, not a permutation


# ---- configs[3]: the 8M population on one GPU ---------------------------------------------------------------------
def test_config4_population_8m_is_step_bit_exact(big, oracle):
    """One IS-only temperature iteration (smc/base.py:401-445 without mutate) over 8M x 32 particles: beta* from the
    device-side search, ESS and evidence ratio, exact-cdf ancestors and the gathered population against the oracle."""
    n, d = 8_000_000, 32
    x, ll, lp, lq = _device_batch(big, n, d, seed=5)
    llh, lph, lqh = (t.cpu().numpy() for t in (ll, lp, lq))
    b, _, conv, _, n_nan, trip, trip_one = big.find_beta(ll, lp, lq, 0.0, 0.5, 1e-6)
    assert conv and n_nan == 0
    ref_b = oracle.determine_beta(llh, lph, lqh, 0.0, beta_tolerance=1e-6, target_efficiency=0.5)
    assert b == ref_b.beta  # same midpoints, same decisions: the same float
    st = smc_math.Stats(*trip, n)
    assert smc_math.ess(st) == pytest.approx(oracle.ess_at_beta(llh, lph, lqh, 0.0, b), rel=1e-9)
    assert smc_math.log_evidence_ratio(st) == pytest.approx(oracle.log_evidence_ratio(llh, lph, lqh, 0.0, b), rel=1e-11)
    idx, _ = smc_math.resample_indices(big, Comm(), ll, lp, lq, 0.0, b, n, np.random.default_rng(8), mode="exact", st=st)
    ref = oracle.resample_indices(llh, lph, lqh, 0.0, b, np.random.default_rng(8).random(n))
    assert np.array_equal(idx.cpu().numpy(), ref)
    xo, llo, lpo, lqo = big.gather(idx, x, ll, lp, lq)
    ref_t = torch.as_tensor(ref, device=big.device)
    assert torch.equal(xo, x[ref_t]) and torch.equal(llo, ll[ref_t]) and torch.equal(lqo, lq[ref_t])
    # the same iteration as ONE chain of launches (asmc_importance_step; 8M: the persistent kernel streams the particles,
    # the tile prefixes come from the separate scan): same beta*, same ancestors
    idx2 = big.importance_step(ll, lp, lq, 0.0, 0.5, 1e-6, smc_math.pcg64_state(np.random.default_rng(8)), n)
    res = big.importance_result()
    assert res[-1] and res[0] == b and res[3] == 3  # rounds: first tree, one prediction window, the final cells
    assert np.array_equal(idx2.cpu().numpy(), ref)
    del xo, x
    torch.cuda.empty_cache()


# ---- configs[2]: 1M x 32, coupling flow on the MFMA + pCN ---------------------------------------------------------
@pytest.fixture(scope="module")
def trained_flow(big):
    from aspire_amd.flows import CouplingFlow

    d = 32
    flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=big.device, dtype=torch.float32, seed=1234)
    flow.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)
    return flow


def test_config3_flow_logprob_1m_vs_oracle_subsample(big, oracle, trained_flow):
    """asmc_coupling_logprob over the whole 1M x 32 batch; a strided 64k subsample against (i) the SAME flow evaluated in
    fp64 - the north star's bar: max relative error <= 1e-6 on these in-distribution rows - and (ii) orc_coupling_logprob
    (fp32 arithmetic in another summation order: two fp32 evaluations, |delta| <= 1e-5 |log q| + 3e-4)."""
    n, d = 1_000_000, 32
    g = torch.Generator(big.device).manual_seed(17)
    x = 1.4 * torch.randn((n, d), device=big.device, dtype=torch.float64, generator=g)
    got = big.coupling_logprob(x, trained_flow.device_coupling(big))
    rows = torch.arange(0, n, n // 65536, device=big.device)[:65536]
    ws, bs = trained_flow.export_layers()
    want = oracle.coupling_logprob(x[rows].cpu().numpy(), ws, bs, trained_flow.loc.cpu().numpy(), trained_flow.scale.cpu().numpy())
    sub = got[rows].cpu().numpy()
    assert np.all(np.isfinite(got.cpu().numpy()))
    np.testing.assert_allclose(sub, want, rtol=1e-5, atol=3e-4)
    from conftest import flow_log_prob_f64

    ref64 = flow_log_prob_f64(trained_flow, x[rows].cpu().numpy())
    rel = np.abs(sub - ref64) / np.maximum(np.abs(ref64), 1.0)
    assert rel.max() <= 1e-6, rel.max()


def test_config3_maf_logprob_1m_vs_fp64_subsample(big):
    """The reference's default flow class (masked autoregressive, flows/torch/flows.py:140-168) at full size: asmc_coupling_logprob
    with kind = ASMC_FLOW_MAF over 1M x 32 rows of a TRAINED flow; a strided 64k subsample against the SAME parameters evaluated
    in fp64 (MAFFlow.log_prob_f64) - the north star's bar: max relative error <= 1e-6 - and every value finite."""
    from aspire_amd.flows import MAFFlow

    n, d = 1_000_000, 32
    flow = MAFFlow(d, n_transforms=3, hidden_features=(64, 64), device=big.device, dtype=torch.float32, seed=1234)
    flow.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)
    g = torch.Generator(big.device).manual_seed(18)
    x = 1.4 * torch.randn((n, d), device=big.device, dtype=torch.float64, generator=g)
    big.profile(True)
    got = big.coupling_logprob(x, flow.device_coupling(big))
    rep = big.profile_report()
    big.profile(False)
    assert any(k.startswith("k_maf_logprob") for k in rep), sorted(rep)
    assert bool(torch.isfinite(got).all())
    rows = torch.arange(0, n, n // 65536, device=big.device)[:65536]
    with torch.no_grad():
        ref64 = flow.log_prob_f64(x[rows]).cpu().numpy()
    sub = got[rows].cpu().numpy()
    rel = np.abs(sub - ref64) / np.maximum(np.abs(ref64), 1.0)
    assert rel.max() <= 1e-6, rel.max()


def test_config3_full_run_1m_d32_flow_pcn_32_steps(big, trained_flow):
    """configs[2] as stated: HipSMC.sample(1M) with the trained coupling-flow proposal, 32 pCN steps per temperature,
    default fp64 noise; the whole mutation loop runs in asmc_pcn_mutate_flow; log Z within 3 sigma of (d/2) log pi."""
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 32, 1 << 20
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=trained_flow, xp=np, engine=big,
                rng=np.random.default_rng(2))
    big.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=32, step_fn="pcn"), store_sample_history=False)
    rep = big.profile_report()
    big.profile(False)
    n_temp = len(sp.history.beta)
    flow_launches = sum(c for k, (c, _) in rep.items() if "coupling" in k or "flow_fused" in k)
    assert flow_launches >= 32 * n_temp and "k_pcn_propose" not in rep  # not the generic split path
    true, err = 0.5 * d * math.log(math.pi), float(out.log_evidence_error)
    assert sp.history.beta[-1] == 1.0 and 0 < err < 0.01
    assert abs(float(out.log_evidence) - true) < 3 * err, (float(out.log_evidence), true, err)
    # Gaussian target, Gaussian reference fitted to the particles: the pCN proposal is (nearly) reversible w.r.t. the
    # tempered target itself, so almost every move is accepted and the step size saturates at its cap
    assert 0.5 < np.mean(sp.history.mcmc_acceptance) <= 1.0
    xs = out.x.double()
    assert float(xs.var(dim=0).mean()) == pytest.approx(0.5, rel=0.02)  # posterior N(0, I/2)
    assert float(xs.mean(dim=0).abs().max()) < 0.01


# ---- configs[4]: 1M x 128 mixture on the fp64 matrix cores --------------------------------------------------------
def _config5_targets(d):
    from aspire_amd.targets import DiagGaussianMixture

    lik = DiagGaussianMixture(np.stack([2 * np.ones(d), -2 * np.ones(d)]), np.stack([0.5 * np.ones(d), np.ones(d)]))
    prior = DiagGaussianMixture.isotropic(d, 0.0, 1.0)

    def lg(mu, var):
        return -0.5 * d * np.log(2 * np.pi * var) - 0.5 * d * mu * mu / var

    # Z = 0.5 N(2 1; 0, (1 + 0.5) I) + 0.5 N(-2 1; 0, (1 + 1) I): Gaussian convolution of each component with the prior
    return lik, prior, float(np.logaddexp(np.log(0.5) + lg(2.0, 1.5), np.log(0.5) + lg(2.0, 2.0)))


def test_config5_pcn_mfma_step_1m_d128_vs_oracle_blocks(big, oracle):
    """k_pcn_mm_* over the whole 1M x 128 population; eight blocks of 256 particles spread over the batch against
    orc_pcn_step with the same global particle ids (the noise streams are keyed by them)."""
    from aspire_amd.flows import GaussianFlow

    n, d, n_steps = 1_000_000, 128, 2
    lik, prior, _ = _config5_targets(d)
    flow = GaussianFlow(d, sigma=3.0, seed=4, engine=big)
    x, lq = flow.sample_and_log_prob(n)
    t_ll, t_lp, t_lq = lik.device_mixture(big), prior.device_mixture(big), flow.device_mixture(big)
    ll, lp = big.mixture_logpdf(x, t_ll), big.mixture_logpdf(x, t_lp)
    g = np.random.default_rng(6)
    a = g.normal(size=(d, d)) / math.sqrt(d)
    cov = 4.0 * (np.eye(d) + 0.3 * (a @ a.T))
    L = np.linalg.cholesky(cov)
    Linv = np.linalg.inv(L)
    mu = 0.2 * g.normal(size=d)
    starts = [0, 4096 * 3 + 16, 131072, 300000 - 7, 500000, 777777, 999000, n - 256]
    keep = [(s, x[s:s + 256].cpu().numpy().copy(), ll[s:s + 256].cpu().numpy().copy(), lp[s:s + 256].cpu().numpy().copy(),
             lq[s:s + 256].cpu().numpy().copy()) for s in starts]
    rho, beta, seed = 0.12, 0.35, 99
    big.profile(True)
    n_acc, _, _ = big.pcn_mutate(x, ll, lp, lq, beta, big.asarray(mu), big.asarray(np.tril(L)), big.asarray(np.tril(Linv)),
                                 t_ll, t_lp, t_lq, seed, 0, rho, n_steps, 3, 0.234, False, "f64")
    rep = big.profile_report()
    big.profile(False)
    assert rep["k_pcn_mm_step"][0] == n_steps and "k_pcn_step_generic" not in rep
    assert 0.02 < n_acc.mean() / n < 0.98
    om = [oracle.Mixture(*(t.cpu().numpy() for t in (m.logw, m.mu, m.prec))) for m in (t_ll, t_lp, t_lq)]
    bad = 0
    for s, xr, llr, lpr, lqr in keep:
        margins = []
        for t in range(n_steps):
            with oracle.accept_margins(256) as m:
                oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, np.tril(L), np.tril(Linv), rho, om[0], om[1], om[2], seed, s, 3 + t, "f64")
            margins.append(m.copy())
        got = x[s:s + 256].cpu().numpy()
        close = np.all(np.abs(got - xr) <= 1e-9 * (1 + np.abs(xr)), axis=1)
        bad += int((~close).sum())
        # a mismatching row sat on a razor edge: its accept margin log_a - log u at some step is at rounding level (the
        # densities here are fp64 on both sides, so the edge is far narrower than with the fp32 flow)
        assert np.all(np.min(np.abs(np.array(margins)), axis=0)[~close] <= 1e-8)
        ok = close
        np.testing.assert_allclose(ll[s:s + 256].cpu().numpy()[ok], llr[ok], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(lq[s:s + 256].cpu().numpy()[ok], lqr[ok], rtol=1e-10, atol=1e-9)
    assert bad <= 2, bad  # razor-edge accept decisions only


@pytest.mark.parametrize("step_fn", ["pcn", "tpcn"])
def test_config5_full_run_1m_d128_mixture(big, step_fn):
    """configs[4] on ONE GPU: 1M x 128, two-component mixture likelihood (examples/smc_example.py lifted to d = 128),
    N(0, I) prior, q = N(0, 3^2 I), adaptive tempering, 32 steps per temperature; log Z against the closed form."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC

    d, n = 128, 1_000_000
    lik, prior, true = _config5_targets(d)
    sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=GaussianFlow(d, sigma=3.0, engine=big, seed=4),
                xp=np, engine=big, rng=np.random.default_rng(1))
    big.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=32, step_fn=step_fn), store_sample_history=False)
    rep = big.profile_report()
    big.profile(False)
    assert rep["k_tpcn_mm_step" if step_fn == "tpcn" else "k_pcn_mm_step"][0] >= 32 * len(sp.history.beta)
    assert "k_pcn_step_generic" not in rep
    err = float(out.log_evidence_error)
    assert sp.history.beta[-1] == 1.0 and 10 < len(sp.history.beta) < 60 and 0 < err < 0.02
    assert abs(float(out.log_evidence) - true) < 3 * err, (float(out.log_evidence), true, err)
    # both modes are populated with the posterior weights of the closed form
    w_plus = float((out.x.double().mean(dim=1) > 0).double().mean())

    def lg(mu, var):
        return -0.5 * d * np.log(2 * np.pi * var) - 0.5 * d * mu * mu / var

    p_plus = float(np.exp(np.log(0.5) + lg(2.0, 1.5) - true))
    assert w_plus == pytest.approx(p_plus, abs=0.01)


# ---- golden G5 with the HIP engine underneath ---------------------------------------------------------------------
def test_draw_initial_samples_golden_through_hip_engine(hip_engine, golden):
    """mcmc.py:49-110 (loop until n valid rows, drop rows with non-finite prior or likelihood, truncate) with the
    compaction done by asmc_compact_valid: the reference's own output (golden ref_initial.npz), bit for bit."""
    from test_host_logic import NumpyGaussFlow, StubSMC

    g = golden["ref_initial"]

    def lp_holes(s):
        x = np.asarray(s.x.cpu() if torch.is_tensor(s.x) else s.x)
        return np.where(x[:, 0] > 1.0, -np.inf, -0.5 * np.sum(x**2, axis=1))

    def ll_holes(s):
        x = np.asarray(s.x.cpu() if torch.is_tensor(s.x) else s.x)
        return np.where(x[:, 1] < -1.5, -np.inf, -0.5 * np.sum(x**2, axis=1))

    sp = StubSMC(log_likelihood=ll_holes, log_prior=lp_holes, dims=3, prior_flow=NumpyGaussFlow(3, 1.0, 41), xp=np,
                 engine=hip_engine)
    hip_engine.profile(True)
    init = sp.draw_initial_samples(500)
    rep = hip_engine.profile_report()
    hip_engine.profile(False)
    assert any("compact" in k for k in rep), rep.keys()  # the HIP compaction ran
    assert np.array_equal(init.x.cpu().numpy(), g["x"])
    assert np.array_equal(init.log_likelihood.cpu().numpy(), g["ll"])
    assert np.array_equal(init.log_prior.cpu().numpy(), g["lp"])
    np.testing.assert_allclose(init.log_q.cpu().numpy(), g["lq"], rtol=1e-15)
    assert sp.n_likelihood_evaluations == int(g["nlike"])


# ---- systematic / stratified resampler (no reference counterpart; the oracle is the contract) ----
@pytest.mark.parametrize("n", [7, 4097, 1_000_000])
@pytest.mark.parametrize("method", ["systematic", "stratified"])
def test_systematic_stratified_indices_bit_exact_vs_oracle(big, oracle, n, method):
    """asmc_systematic_uniforms + asmc_search against orc_systematic_uniforms / orc_stratified_uniforms +
    searchsorted(right) on the oracle's sequential cdf (oracle/asmc_oracle.c:376-386): uniforms and ancestors bit-exact,
    for n_out = N, a ragged n_out != N, and with the output slots split at sharded j0 offsets."""
    _, ll, lp, lq = synth(n, 2, 500 + n % 97)
    lld, lpd, lqd = dev(big, ll, lp, lq)
    beta = 0.05
    for n_out in (n, max(1, (3 * n) // 5 + 1)):
        rng = np.random.default_rng(77)
        if method == "systematic":
            u_ref = oracle.systematic_uniforms(n_out, float(rng.random()))
        else:
            u_ref = oracle.stratified_uniforms(rng.random(n_out))
        ref, cdf = oracle.resample_indices(ll, lp, lq, 0.0, beta, u_ref, return_cdf=True)
        idx, j0 = smc_math.resample_indices(big, Comm(), lld, lpd, lqd, 0.0, beta, n_out, np.random.default_rng(77),
                                            mode="exact", method=method)
        assert j0 == 0 and np.array_equal(idx.cpu().numpy(), ref)
        assert np.all(np.diff(ref) >= 0)  # sorted draws give sorted ancestors
        # sharded output slots: rank r of 3 fills [j0, j1) from the same u0 / the same stratified stream
        got_u = []
        for r in range(3):
            per = -(-n_out // 3)
            a, b = min(r * per, n_out), min(r * per + per, n_out)
            u = smc_math.draw_uniforms(big, np.random.default_rng(77), n_out, a, b - a, method)
            got_u.append(u.cpu().numpy())
        assert np.array_equal(np.concatenate(got_u), u_ref)
    # the device cdf the searches ran on: numpy's cumsum of the device's own weights bit for bit, and the oracle's cdf to
    # rounding (the second log-sum-exp is summed in a different order on the two sides: a common factor of 1 +- 1e-16)
    w = big.normalized_weights(lld, lpd, lqd, 0.0, beta, *_shift_lse(big, lld, lpd, lqd, beta, n))
    cdf_dev, _ = big.cdf(w, "exact", 0.0, want_total=False, normalize=True)
    cs = np.cumsum(w.cpu().numpy())
    assert np.array_equal(cdf_dev.cpu().numpy(), cs / cs[-1])
    np.testing.assert_allclose(cdf_dev.cpu().numpy(), cdf, rtol=1e-14)


def _shift_lse(eng, ll, lp, lq, beta, n):
    st = smc_math.global_stats(eng, Comm(), ll, lp, lq, 0.0, [beta], n)[0]
    shift = float((st.m + np.log(st.S1)) - math.log(n))
    mp = st.m + shift
    s1p = eng.weights_sums(ll, lp, lq, 0.0, [beta], [mp], [shift])[0, 0]
    return shift, float(mp + np.log(s1p))


@pytest.mark.parametrize("method", ["systematic", "stratified"])
def test_sampler_run_with_systematic_resampling_gets_the_evidence(big, method):
    """A full sampler run with resample_method = systematic / stratified: log Z against (d/2) log pi."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 16, 1 << 18
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, engine=big, seed=9),
                xp=np, engine=big, rng=np.random.default_rng(4))
    out = sp.sample(n, sampler_kwargs=dict(n_steps=16, step_fn="pcn"), store_sample_history=False, resample_method=method)
    true, err = 0.5 * d * math.log(math.pi), float(out.log_evidence_error)
    assert sp.history.beta[-1] == 1.0
    assert abs(float(out.log_evidence) - true) < 3 * err + 2e-3, (float(out.log_evidence), true, err)
    assert float(out.x.double().var(dim=0).mean()) == pytest.approx(0.5, rel=0.03)


# ---- checkpoint state pinned to the reference, with the HIP engine underneath ----------------------
def test_checkpoint_states_match_reference_golden_hip(hip_engine, golden):
    """The reference's own checkpoint dictionaries (golden ref_checkpoint.npz: keys, types, beta, iteration, generator
    state, history, particles) reproduced by the loop running on the HIP kernels; samples arrive as host arrays."""
    from test_checkpoint import check_states_match_reference, run_with_checkpoints

    sp, out, states = run_with_checkpoints(hip_engine)
    check_states_match_reference(states, golden["ref_checkpoint"])


def test_resume_from_reference_state_on_hip_engine(hip_engine, golden):
    """Resume from the REFERENCE's mid-run state (golden data): the GPU loop continues the reference's beta schedule and
    ends on the reference's particles."""
    from test_checkpoint import check_resume_continues_reference_schedule

    check_resume_continues_reference_schedule(hip_engine, golden["ref_checkpoint"])
