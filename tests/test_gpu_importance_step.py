"""asmc_importance_step (search + evidence moments + resampling as one chain of launches, one synchronisation) against
the step-by-step entry points it fuses, against the oracle, and inside the sampler loop.

Bars: beta* and the resampling indices bit-exact; the reduced sums within 1e-12 (they are summed in another order)."""
import math

import numpy as np
import pytest
import torch

from conftest import synth

from aspire_amd import smc_math
from aspire_amd.comm import Comm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(hip_engine):
    return hip_engine


def dev(eng, *arrs):
    return tuple(eng.asarray(a) for a in arrs)


def _step_by_step(eng, lld, lpd, lqd, beta0, target, tol, n, seed):
    b, eff1, conv, passes, n_nan, trip, trip_one = eng.find_beta(lld, lpd, lqd, beta0, target, tol)
    assert conv and n_nan == 0 and trip is not None
    st = smc_math.Stats(*trip, n)
    var, s1p = smc_math.evidence_variance_and_lse(eng, Comm(), lld, lpd, lqd, beta0, b, st)
    rng = np.random.default_rng(seed)
    idx, _ = smc_math.resample_indices(eng, Comm(), lld, lpd, lqd, beta0, b, n, rng, mode="exact", st=st, s1p=s1p)
    return b, eff1, passes, trip, trip_one, var, s1p, idx.cpu().numpy(), rng


@pytest.mark.parametrize("n,d,seed,beta0,target", [
    (3, 2, 1, 0.0, 0.5), (64, 2, 2, 0.0, 0.7), (257, 2, 2, 0.2, 0.5), (1000, 2, 3, 0.0, 0.5), (4097, 3, 4, 0.0, 0.5), (300001, 4, 5, 0.1, 0.6), (1 << 20, 4, 6, 0.0, 0.5),
    (1 << 20, 2, 7, 0.25, 0.9), (1_500_003, 2, 8, 0.0, 0.5), (2_000_000, 2, 9, 0.3, 0.3)])
def test_importance_step_equals_the_step_by_step_path(eng, n, d, seed, beta0, target):
    """n <= 1M: the persistent kernel keeps the particles in LDS; above it streams them once per phase."""
    x, ll, lp, lq = synth(n, d, seed)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    tol = 1e-6
    b, eff1, passes, trip, trip_one, var, s1p, idx_ref, rng_ref = _step_by_step(eng, lld, lpd, lqd, beta0, target, tol, n, 21)
    rng = np.random.default_rng(21)
    idx = eng.importance_step(lld, lpd, lqd, beta0, target, tol, smc_math.pcg64_state(rng), n)
    fb, feff1, fconv, fpasses, fnan, ftrip, ftrip_one, m2, fs1p, found = eng.importance_result()
    assert found and fconv and fnan == 0
    assert fb == b and fpasses == passes  # same candidates, same decisions
    assert feff1 == pytest.approx(eff1, rel=1e-12)
    assert ftrip[0] == trip[0] and ftrip_one[0] == trip_one[0]  # shifts / maxima: exact
    np.testing.assert_allclose(ftrip[1:], trip[1:], rtol=1e-12)
    np.testing.assert_allclose(ftrip_one[1:], trip_one[1:], rtol=1e-12)
    st = smc_math.Stats(*ftrip, n)
    mean_u = st.S1 / n
    assert (m2 / n) / (n * mean_u**2) == pytest.approx(var, rel=1e-10)
    assert fs1p == pytest.approx(s1p, rel=1e-12)
    got = idx.cpu().numpy()
    assert np.array_equal(got, idx_ref), (int((got != idx_ref).sum()), n)


def test_importance_step_indices_vs_oracle_1m(eng, oracle):
    """Against numpy's algorithm itself (oracle: sequential cumsum + searchsorted on the host uniforms)."""
    n = 1 << 20
    x, ll, lp, lq = synth(n, 2, 99)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    rng = np.random.default_rng(5)
    idx = eng.importance_step(lld, lpd, lqd, 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
    b, _, conv, _, _, trip, _, _, _, found = eng.importance_result()
    assert found and conv
    ref = oracle.resample_indices(ll, lp, lq, 0.0, b, np.random.default_rng(5).random(n))
    assert np.array_equal(idx.cpu().numpy(), ref)
    # the oracle's own search lands on the same beta*
    target = 0.5
    assert oracle.ess_at_beta(ll, lp, lq, 0.0, b) / n >= target
    assert oracle.ess_at_beta(ll, lp, lq, 0.0, min(1.0, b + 2e-6)) / n < target


@pytest.mark.parametrize("n_out", [1, 63, 1000, 262144, 262145, 1_048_577])
def test_importance_step_draw_counts(eng, n_out):
    """n_out != N: the search's thread count and strides adapt; still the stream's first n_out doubles."""
    n = 200_003
    x, ll, lp, lq = synth(n, 2, 31)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    rng = np.random.default_rng(8)
    idx = eng.importance_step(lld, lpd, lqd, 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n_out)
    b, _, conv, _, _, trip, _, _, s1p, found = eng.importance_result()
    assert found
    ref, _ = smc_math.resample_indices(eng, Comm(), lld, lpd, lqd, 0.0, b, n_out, np.random.default_rng(8), mode="exact",
                                       st=smc_math.Stats(*trip, n))
    assert np.array_equal(idx.cpu().numpy(), ref.cpu().numpy())


def test_importance_step_heavy_weights_guide_table(eng):
    """A few particles own most of the mass: long runs of guide buckets per element (the block-wide fill)."""
    n = 1 << 19
    g = np.random.default_rng(17)
    ll = g.normal(size=n)
    ll[[5, 70000, n - 3]] += 14.0  # three particles carry ~all of the weight at beta = 1
    lp, lq = g.normal(size=n) * 0.01, g.normal(size=n) * 0.01
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    for target in (0.5, 1e-6):  # 1e-6: eff(1) >= target -> beta* = 1 with three dominant weights
        b, eff1, passes, trip, trip_one, var, s1p, idx_ref, _ = _step_by_step(eng, lld, lpd, lqd, 0.0, target, 1e-6, n, 4)
        rng = np.random.default_rng(4)
        idx = eng.importance_step(lld, lpd, lqd, 0.0, target, 1e-6, smc_math.pcg64_state(rng), n)
        res = eng.importance_result()
        assert res[-1] and res[0] == b
        assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert b == 1.0


def test_importance_step_reports_failures(eng):
    """NaN log-weights and a search that cannot leave beta0 are reported, and the parked draw stays in bounds."""
    n = 100_000
    x, ll, lp, lq = synth(n, 2, 41)
    ll_nan = ll.copy()
    ll_nan[123] = np.nan
    rng = np.random.default_rng(1)
    idx = eng.importance_step(*dev(eng, ll_nan, lp, lq), 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
    res = eng.importance_result()
    assert res[4] == 1 and not res[-1]
    got = idx.cpu().numpy()
    assert got.min() >= 0 and got.max() <= n
    # target efficiency above 1: no beta > beta0 qualifies
    idx = eng.importance_step(*dev(eng, ll, lp, lq), 0.2, 1.5, 1e-6, smc_math.pcg64_state(rng), n)
    res = eng.importance_result()
    assert not res[-1] and res[5] is None
    # back-to-back launches share the barrier counter: a normal step afterwards still works
    idx = eng.importance_step(*dev(eng, ll, lp, lq), 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
    assert eng.importance_result()[-1]


def _np(t):
    return t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _run(eng, fused, n, d, seed, **kw):
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, engine=eng, seed=1),
                xp=np, engine=eng, rng=np.random.default_rng(seed))
    sp.fused_importance_step = fused
    out = sp.sample(n, sampler_kwargs=dict(n_steps=4), store_sample_history=False, **kw)
    return sp, out


@pytest.mark.parametrize("n,d", [(20000, 8), (300000, 32)])
def test_sampler_with_and_without_the_fused_step(eng, n, d):
    """Same beta schedule, same particles, same generator state; evidence terms to rounding."""
    sp_a, out_a = _run(eng, True, n, d, 3)
    sp_b, out_b = _run(eng, False, n, d, 3)
    assert sp_a.history.beta == sp_b.history.beta
    np.testing.assert_allclose(sp_a.history.log_norm_ratio, sp_b.history.log_norm_ratio, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(sp_a.history.log_norm_ratio_var, sp_b.history.log_norm_ratio_var, rtol=1e-9)
    np.testing.assert_allclose(sp_a.history.ess, sp_b.history.ess, rtol=1e-11)
    assert np.array_equal(_np(out_a.x), _np(out_b.x))
    assert sp_a.rng.bit_generator.state == sp_b.rng.bit_generator.state
    d_ = d
    assert abs(float(out_a.log_evidence) - 0.5 * d_ * math.log(math.pi)) < max(5 * float(out_a.log_evidence_error), 0.05)


def test_sampler_fused_step_is_used_and_schedule_clamps_fall_back(eng):
    """The fused chain is what runs (k_is_weights / k_search_pcg launches, no k_bis_sums); with a max beta step that
    caps beta below beta* the parked rows are dropped and the result equals the step-by-step run."""
    eng.profile(True)
    sp, out = _run(eng, True, 50000, 8, 5)
    rep = eng.profile_report()
    eng.profile(False)
    temps = len(sp.history.beta)
    assert rep["k_is_weights"][0] == temps
    assert "k_bis_sums" not in rep and "k_search_pcg" in rep
    sp_a, out_a = _run(eng, True, 50000, 8, 6, max_beta_step=0.05)
    sp_b, out_b = _run(eng, False, 50000, 8, 6, max_beta_step=0.05)
    assert sp_a.history.beta == sp_b.history.beta and max(np.diff([0.0] + sp_a.history.beta)) <= 0.05 + 1e-12
    assert np.array_equal(_np(out_a.x), _np(out_b.x))
    assert sp_a.rng.bit_generator.state == sp_b.rng.bit_generator.state


def test_reference_fit_moments_ride_on_the_fused_step(eng, monkeypatch):
    """`pcn` steps fit their reference Gaussian to the moments of the resampled population: the fused importance step
    enqueues them behind its gather (asmc_mean_gram_enqueue / _fetch) and the mutation finds them parked - one moments
    pass per temperature, no pass of its own - with results identical to the run that computes them in the mutation."""
    from aspire_amd.samplers.smc import HipSMC

    kw = dict(n_steps=3, step_fn="pcn")

    def run(seed):
        from aspire_amd.flows import GaussianFlow
        from aspire_amd.targets import DiagGaussianMixture

        lik = DiagGaussianMixture.isotropic(32, normalized=False)
        sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=32, prior_flow=GaussianFlow(32, sigma=1.5, engine=eng, seed=1),
                    xp=np, engine=eng, rng=np.random.default_rng(seed))
        eng.profile(True)
        out = sp.sample(100_000, sampler_kwargs=dict(kw), store_sample_history=False)
        rep = eng.profile_report()
        eng.profile(False)
        return sp, out, rep

    sp_a, out_a, rep_a = run(4)
    temps = len(sp_a.history.beta)
    # (the column sums ride along the gather: no k_colsum pass over the rows it has just written)
    assert rep_a["k_is_weights"][0] == temps and "k_colsum<double>" not in rep_a and rep_a["k_gram_mm"][0] == temps
    monkeypatch.setattr(HipSMC, "_speculated_moments_n", lambda self, samples: None)
    sp_b, out_b, rep_b = run(4)
    assert rep_b["k_gram_mm"][0] == temps and rep_b["k_reduce_columns"][0] == temps
    # (a fit behind the gather starts from the column sums that rode along it, a fit at mutation time from its own pass over the
    # rows: the two reference Gaussians agree to rounding, so the runs agree up to accept decisions at a razor's edge)
    assert len(sp_a.history.beta) == len(sp_b.history.beta) and sp_a.history.beta[0] == sp_b.history.beta[0]
    np.testing.assert_allclose(sp_a.history.beta, sp_b.history.beta, rtol=1e-4)
    np.testing.assert_allclose(sp_a.history.mcmc_acceptance, sp_b.history.mcmc_acceptance, atol=2e-3)
    same = np.all(np.abs(_np(out_a.x) - _np(out_b.x)) <= 1e-9 * (1 + np.abs(_np(out_b.x))), axis=1)
    assert same.mean() > 0.5
    assert abs(float(out_a.log_evidence) - float(out_b.log_evidence)) < 3 * float(out_a.log_evidence_error)
    # EXACT leg (round-3 advice): with the ride-along column sums switched off (ASMC_GATHER_NO_COLSUM=1) both arms start their
    # reference fit from the same k_colsum pass, so a fit enqueued behind the gather and a fit at mutation time must agree bit
    # for bit - stale moments or wrong rows in the ahead path cannot hide behind the loose comparison above
    monkeypatch.undo()
    monkeypatch.setenv("ASMC_GATHER_NO_COLSUM", "1")
    sp_c, out_c, rep_c = run(4)
    assert rep_c["k_is_weights"][0] == temps and rep_c["k_gram_mm"][0] == temps
    monkeypatch.setattr(HipSMC, "_speculated_moments_n", lambda self, samples: None)
    sp_d, out_d, rep_d = run(4)
    assert sp_c.history.beta == sp_d.history.beta and sp_c.history.mcmc_acceptance == sp_d.history.mcmc_acceptance
    assert sp_c.history.log_norm_ratio == sp_d.history.log_norm_ratio
    assert float(out_c.log_evidence) == float(out_d.log_evidence) and np.array_equal(_np(out_c.x), _np(out_d.x))
    assert sp_c.rng.bit_generator.state == sp_d.rng.bit_generator.state


def test_importance_step_enqueued_behind_the_mutation(eng, monkeypatch):
    """Flow-proposal runs put the next temperature's importance step onto the stream behind the mutation, before the host
    waits for either (asmc_pcn_mutate_flow_enqueue / _result: the result waits for an event behind the mutation's own
    read-back, the step's results are collected at the top of the next iteration).  Same schedule, same log Z, same particles
    as the run that starts the step after the mutation has returned (ASMC_IS_AHEAD=0), and the path is the one taken."""
    from conftest import random_coupling_flow

    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d = 32
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    flow = random_coupling_flow(d, 4, 64)
    taken = []
    orig = eng.pcn_mutate_flow_enqueue

    def run():
        flow._draws = flow._hip_draws = 0
        sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(9),
                    dtype="float64")
        out = sp.sample(60_000, sampler_kwargs=dict(n_steps=3, step_fn="pcn"), store_sample_history=False)
        return sp, out

    monkeypatch.setattr(eng, "pcn_mutate_flow_enqueue", lambda *a, **k: (taken.append(1), orig(*a, **k))[1])
    sp_a, out_a = run()
    assert len(taken) == len(sp_a.history.beta) - 1  # every mutation but the last one has a step behind it
    monkeypatch.setenv("ASMC_IS_AHEAD", "0")
    n_before = len(taken)
    sp_b, out_b = run()
    assert len(taken) == n_before
    assert sp_a.history.beta == sp_b.history.beta and sp_a.history.mcmc_acceptance == sp_b.history.mcmc_acceptance
    assert sp_a.history.log_norm_ratio == sp_b.history.log_norm_ratio
    assert float(out_a.log_evidence) == float(out_b.log_evidence) and np.array_equal(_np(out_a.x), _np(out_b.x))
    assert sp_a.rng.bit_generator.state == sp_b.rng.bit_generator.state


_CKPT = []


def _ckpt_cb(state):  # module level: the checkpoint state carries its callback, and local functions do not pickle
    import pickle

    _CKPT.append(pickle.dumps(state))


def test_pending_importance_step_survives_history_and_checkpoints(eng, monkeypatch):
    """Between the mutation that enqueued the next importance step and the iteration that collects it, the sampler stores the
    sample history and calls the checkpoint callback: the pending step must not leak into either (pickled states drop it),
    and the run equals the one without the look-ahead."""
    from conftest import random_coupling_flow

    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d = 32
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    flow = random_coupling_flow(d, 4, 64)

    def run(ahead):
        monkeypatch.setenv("ASMC_IS_AHEAD", ahead)
        flow._draws = flow._hip_draws = 0
        _CKPT.clear()
        sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(9),
                    dtype="float64")
        out = sp.sample(50_000, sampler_kwargs=dict(n_steps=3, step_fn="pcn"), store_sample_history=True,
                        checkpoint_callback=_ckpt_cb, checkpoint_every=1)
        return sp, out, list(_CKPT)

    sp_a, out_a, st_a = run("1")
    sp_b, out_b, st_b = run("0")
    assert sp_a.history.beta == sp_b.history.beta and float(out_a.log_evidence) == float(out_b.log_evidence)
    assert len(sp_a.history.sample_history) == len(sp_a.history.beta) + 1 and len(st_a) == len(st_b) > 0
    assert all(np.array_equal(np.asarray(a.x), np.asarray(b.x))
               for a, b in zip(sp_a.history.sample_history, sp_b.history.sample_history))
    assert [len(a) for a in st_a] == [len(b) for b in st_b]  # no device rows, no pending step inside the pickles


def test_barrier_timeout_abandons_the_step_and_the_sampler_falls_back(monkeypatch):
    """A launch of the persistent kernel that is not fully resident (two such kernels of different processes sharing the
    GPU) must not hang: its barriers time out, the step reports found = 0 with in-bounds indices, the context stops
    using the kernel, and the sampler carries on through the step-by-step entry points with the same results."""
    from aspire_amd.engine import HipEngine

    e2 = HipEngine(0, n_max=1 << 17, d_max=8)  # its own context: the fallback is sticky
    n = 100_000
    x, ll, lp, lq = synth(n, 2, 41)
    rng = np.random.default_rng(1)
    monkeypatch.setenv("ASMC_ISW_TEST_TIMEOUT", "1")
    idx = e2.importance_step(*(e2.asarray(a) for a in (ll, lp, lq)), 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
    res = e2.importance_result()
    monkeypatch.delenv("ASMC_ISW_TEST_TIMEOUT")
    assert not res[-1] and not res[2] and e2.importance_step_disabled
    got = idx.cpu().numpy()
    assert got.min() >= 0 and got.max() < n
    with pytest.raises(Exception):
        e2.importance_step(*(e2.asarray(a) for a in (ll, lp, lq)), 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
    sp_a, out_a = _run(e2, True, 20000, 8, 3)   # speculation is skipped on this engine now
    sp_b, out_b = _run(e2, False, 20000, 8, 3)
    assert sp_a.history.beta == sp_b.history.beta and np.array_equal(_np(out_a.x), _np(out_b.x))


@pytest.mark.parametrize("n", [200_000, 1_300_000])
def test_gather_reuses_the_records_the_step_packed_and_only_those(eng, n):
    """k_is_weights writes the (ll, lp, lq, 0) records of asmc_gather on its way (resident and streaming variants): a gather
    of the same arrays right behind the step launches no packing pass; any library launch in between, or other arrays, and
    the gather packs again.  Rows and scalars equal torch indexing every time."""
    x, ll, lp, lq = synth(n, 4, 31)
    xd, lld, lpd, lqd = dev(eng, x, ll, lp, lq)
    rng = np.random.default_rng(5)

    def gathered(call_between, ll_arg):
        idx = eng.importance_step(lld, lpd, lqd, 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
        if call_between:
            eng.count_nonfinite(lld)
        eng.profile(True)
        rows = eng.gather(idx, xd, ll_arg, lpd, lqd)
        rep = eng.profile_report()
        eng.profile(False)
        eng.importance_result()
        assert torch.equal(rows[0], xd[idx]) and torch.equal(rows[1], ll_arg[idx])
        assert torch.equal(rows[2], lpd[idx]) and torch.equal(rows[3], lqd[idx])
        return "k_pack_records" in rep

    assert not gathered(False, lld)
    assert gathered(True, lld)
    assert gathered(False, lld.clone())


@pytest.mark.parametrize("d", [32, 64, 16])
def test_column_sums_ride_along_the_gather_and_only_for_its_rows(eng, d):
    """asmc_gather of fp64 rows leaves the column-sum partials of the rows it wrote; asmc_mean_gram_enqueue / asmc_colsum_dev of
    exactly those rows, with no other launch in between, start from them (no k_colsum pass) and agree with the column sums of a
    separate pass to rounding; any launch in between, or other rows, and the pass runs."""
    n = 150_000
    g = torch.Generator(eng.device).manual_seed(d)
    x = (0.2 + torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)).contiguous()
    ll = torch.randn(n, device=eng.device, dtype=torch.float64, generator=g)
    idx = torch.randint(0, n, (n,), device=eng.device, generator=g)

    def sums(call_between, other_rows):
        rows = eng.gather(idx, x, ll, ll, ll)
        if call_between:
            eng.count_nonfinite(ll)
        target = rows[0].clone() if other_rows else rows[0]
        eng.profile(True)
        s_d = eng.colsum_dev(target, gathered=True)
        rep = eng.profile_report()
        eng.profile(False)
        ref = eng.colsum(target)
        np.testing.assert_allclose(s_d.cpu().numpy(), ref, rtol=1e-12, atol=1e-9)
        assert torch.equal(rows[0], x[idx])
        return "k_colsum<double>" in rep

    assert not sums(False, False)
    assert sums(True, False) and sums(False, True)
    rows = eng.gather(idx, x, ll, ll, ll)  # without the caller's word the pass runs (the library cannot see a caller's own kernels)
    eng.profile(True)
    eng.colsum_dev(rows[0])
    assert "k_colsum<double>" in eng.profile_report()
    eng.profile(False)
    if d in (32, 64):  # the enqueue form behind the importance step's gather
        rows = eng.gather(idx, x, ll, ll, ll)
        eng.profile(True)
        assert eng.mean_gram_enqueue(rows[0], n, gathered=True)
        rep = eng.profile_report()
        eng.profile(False)
        s, gr = eng.mean_gram_fetch(d)
        assert "k_colsum<double>" not in rep and "k_gram_mm" in rep
        np.testing.assert_allclose(s, eng.colsum(rows[0]), rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(gr, eng.centered_gram(rows[0], s / n), rtol=1e-9, atol=1e-6)
