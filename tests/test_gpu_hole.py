"""Non-finite log-targets inside the mutation, on the HIP engine (round 6).

The reference's own integration test (`/root/reference/tests/integration_tests/test_integration.py:131-166`) punches a hole of
-inf, NaN or +inf into the likelihood and expects `sample_posterior(sampler="smc")` to finish.  NaN -> -inf is the reference's
rule (`smc/base.py:518`); +inf -> -inf is this repository's reading of what the third-party step (minipcn, absent) must do for
that test to pass: a proposal whose tempered log-target is +inf is rejected and counted as a rejection, so that no particle ever
carries `log_likelihood = +inf` into the next log-sum-exp (`csrc/asmc_pcn_dev.h` `log_p_t`, `oracle/asmc_oracle.c`
`orc_log_p_t`).  Here:
* the reference's test itself through `aspire_amd.Aspire` with Python callables (numpy and torch) on the HIP engine;
* the accept kernel of the split path with non-finite proposed values, against the rule itself;
* every one-kernel step family (register-resident d <= 32, fp64-MFMA d = 64 / 128, fused flow d = 32, flow16 d = 64) with
  CARRIED non-finite values (what a caller-supplied state may hold): decisions against the oracle's restatement of the step.
tests/test_likelihood_hole.py is the CPU twin on the test double.
"""
import numpy as np
import pytest
import torch

from test_likelihood_hole import HOLE_VALUES, IDS, check_hole_run, run_hole

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(hip_engine):
    return hip_engine


@pytest.mark.parametrize("value", HOLE_VALUES, ids=IDS)
def test_smc_log_likelihood_with_invalid_value_python_callables(eng, value):
    """test_integration.py:131-166 with numpy callables (the split path: propose kernel -> callables -> accept kernel)."""
    n = 4000
    asp, out, history = run_hole(value, eng, n, seed=21)
    check_hole_run(asp, out, history, n)
    assert "split" in str(asp.sampler.last_mutation_path) or "callable" in str(asp.sampler.last_mutation_path), asp.sampler.last_mutation_path


@pytest.mark.parametrize("value", HOLE_VALUES, ids=IDS)
def test_smc_log_likelihood_with_invalid_value_default_sampler_kwargs_small(eng, value):
    """... and at the reference test's own size (n_samples = 100, every default)."""
    asp, out, history = run_hole(value, eng, 100, seed=5)
    check_hole_run(asp, out, history, 100, sigmas=4.0, slack=0.1)


@pytest.mark.parametrize("value", HOLE_VALUES, ids=IDS)
def test_split_accept_kernel_rejects_non_finite_proposals(eng, oracle, value):
    """k_pcn_accept_flags: proposed (ll', lp', lq') with a non-finite entry - with and without a log-Jacobian term - is never
    accepted; rows whose proposal is finite decide exactly as the rule `log u < log_a` says; a carried non-finite tempered
    log-target (-> -inf) loses against any finite proposal."""
    n, d, beta = 4096, 8, 0.35
    g = np.random.default_rng(3)
    x = g.normal(size=(n, d))
    xp_ = x + 0.1 * g.normal(size=(n, d))
    ll, lp, lq = (g.normal(size=n) for _ in range(3))
    lln, lpn, lqn = (g.normal(size=n) for _ in range(3))
    q0, q1 = g.random(n), g.random(n)
    bad_new, bad_old = np.arange(n) % 7 == 0, np.arange(n) % 11 == 3
    lln[bad_new] = value
    lqn[(np.arange(n) % 13 == 5)] = value
    bad_new |= np.arange(n) % 13 == 5
    ll[bad_old] = value
    for with_lj in (False, True):
        ljo, ljn = (g.normal(size=n), g.normal(size=n)) if with_lj else (None, None)
        xd, xpd = eng.asarray(x), eng.asarray(xp_)
        lld, lpd, lqd = eng.asarray(ll), eng.asarray(lp), eng.asarray(lq)
        args = [eng.asarray(a) for a in (lln, lpn, lqn, q0, q1)]
        ljod, ljnd = (eng.asarray(ljo), eng.asarray(ljn)) if with_lj else (None, None)
        nacc = eng.pcn_accept(xd, xpd, lld, lpd, lqd, args[0], args[1], args[2], args[3], args[4], beta, 77, 5, 2, ljod, ljnd)
        moved = (xd != eng.asarray(x)).any(dim=1).cpu().numpy()
        assert not moved[bad_new].any()  # a non-finite proposal never wins
        new = np.array([oracle.log_p_t(a, b, c, beta) for a, b, c in zip(lln, lpn, lqn)])
        old = np.array([oracle.log_p_t(a, b, c, beta) for a, b, c in zip(ll, lp, lq)])
        if with_lj:
            with np.errstate(all="ignore"):
                new, old = new + ljn, old + ljo
            new, old = np.where(new < np.inf, new, -np.inf), np.where(old < np.inf, old, -np.inf)
        u = np.array([oracle.pcn_noise(77, 5 + i, 2, d)[1] for i in range(n)])
        with np.errstate(all="ignore"):
            expect = np.log(u) < (new + 0.5 * q1) - (old + 0.5 * q0)
        assert np.array_equal(moved, expect) and nacc == int(expect.sum())
        assert expect[bad_old & ~bad_new].all()  # carried non-finite: any finite proposal is taken
        got_ll = lld.cpu().numpy()
        assert np.isfinite(got_ll[moved]).all() and np.array_equal(got_ll[moved], lln[moved])


def _gauss(d, g, scale=1.0):
    v = g.uniform(0.7, 1.5, size=(1, d)) * scale
    return ([-0.5 * d * np.log(2 * np.pi) - 0.5 * np.log(v).sum()], 0.2 * g.normal(size=(1, d)), 1 / v)


@pytest.mark.parametrize("d,nu", [(32, 0.0), (8, 5.0), (64, 0.0), (128, 4.0), (20, 0.0)])
def test_builtin_target_steps_with_carried_non_finite_values_vs_oracle(eng, oracle, d, nu):
    """k_pcn_reg (d <= 32) / k_pcn_mm (d = 64 / 128), pCN and tpCN, built-in densities: rows that CARRY ll = +inf, NaN or -inf
    (their tempered log-target is -inf under the rule) take their first proposal, exactly as the oracle's restatement decides;
    afterwards every carried value is the density at the stored position again."""
    n, beta, rho = 3000, 0.45, 0.3
    g = np.random.default_rng(100 + d)
    x = 0.9 * g.normal(size=(n, d))
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.tril(np.linalg.cholesky(0.8 * (np.eye(d) + 0.2 * a @ a.T)))
    Linv, mu = np.tril(np.linalg.inv(L)), 0.05 * g.normal(size=d)
    mixes = [_gauss(d, g), _gauss(d, g), _gauss(d, g, 2.25)]
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    ll[0::9], ll[1::9], lq[2::9], lp[3::9] = np.inf, np.nan, np.inf, -np.inf
    special = np.zeros(n, bool)
    for k in range(4):
        special[k::9] = True
    xd, lld, lpd, lqd = (eng.asarray(v) for v in (x, ll, lp, lq))
    n_acc, _, _ = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0], dm[1], dm[2],
                                 4242, 17, rho, 1, 3, 0.234, False, "f64", nu)
    xr, llr, lpr, lqr = x.copy(), ll.copy(), lp.copy(), lq.copy()
    if nu > 0:
        acc_ref = oracle.tpcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, om[0], om[1], om[2], 4242, 17, 3)
    else:
        acc_ref = oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, om[0], om[1], om[2], 4242, 17, 3)
    got = xd.cpu().numpy()
    moved_g = np.any(np.abs(got - x) > 1e-9 * (1 + np.abs(x)), axis=1)
    moved_r = np.any(xr != x, axis=1)
    assert moved_r[special].all() and moved_g[special].all()  # -inf carried log-target: the (finite) proposal always wins
    assert (moved_g != moved_r).sum() <= 2 and abs(int(n_acc[0]) - acc_ref) <= 2
    same = moved_g == moved_r
    np.testing.assert_allclose(got[same], xr[same], rtol=1e-9, atol=1e-10)
    for dv, m in ((lld, om[0]), (lpd, om[1]), (lqd, om[2])):
        v = dv.cpu().numpy()
        assert np.isfinite(v[special]).all()
        np.testing.assert_allclose(v[moved_g], m.logpdf(got)[moved_g], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("kind,d,nu", [("coupling", 32, 0.0), ("coupling", 32, 5.0), ("maf", 32, 0.0), ("coupling", 64, 0.0), ("maf", 64, 4.0),
                                       ("coupling", 128, 0.0)])
def test_flow_proposal_steps_with_carried_non_finite_values_vs_oracle(eng, oracle, kind, d, nu):
    """k_pcn_flow_fused (d = 32) and k_pcn_flow16 / k_tpcn_flow16 (d = 64 / 128): the same, with a neural proposal density."""
    from conftest import random_coupling_flow, random_maf_flow

    n, beta, rho = 3000, 0.4, 0.3
    flow = random_coupling_flow(d, 3, 64, seed=6) if kind == "coupling" else random_maf_flow(d, 3, 64, seed=6)
    dev = flow.device_coupling(eng)
    ws, bs = flow.export_layers()
    g = np.random.default_rng(200 + d)
    x = 0.9 * g.normal(size=(n, d))
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.tril(np.linalg.cholesky(0.8 * (np.eye(d) + 0.2 * a @ a.T)))
    Linv, mu = np.tril(np.linalg.inv(L)), 0.05 * g.normal(size=d)
    m = ([0.0], np.zeros((1, d)), np.ones((1, d)))
    tgt_o, t_ll = oracle.Mixture(*m), eng.make_mixture(*m)
    flp = oracle.coupling_logprob if kind == "coupling" else oracle.maf_logprob
    ll = tgt_o.logpdf(x)
    lp, lq = ll.copy(), flp(x, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    ll[0::9], ll[1::9], lq[2::9], lp[3::9] = np.inf, np.nan, np.inf, -np.inf
    special = np.zeros(n, bool)
    for k in range(4):
        special[k::9] = True
    xd, lld, lpd, lqd = (eng.asarray(v) for v in (x, ll, lp, lq))
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate_flow(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), t_ll, t_ll, dev,
                                      4242, 17, rho, 1, 9, 0.234, False, "f64", nu)
    rep = eng.profile_report()
    eng.profile(False)
    step_k = "k_pcn_flow_fused" if d == 32 else ("k_tpcn_flow16" if nu > 0 else "k_pcn_flow16")
    assert rep[step_k][0] == 1, sorted(rep)
    xr, llr, lpr, lqr = x.copy(), ll.copy(), lp.copy(), lq.copy()
    with oracle.accept_margins(n) as mg:
        if nu > 0:
            acc_ref = oracle.tpcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, tgt_o, tgt_o, ws, bs, flow.loc.numpy(),
                                            flow.scale.numpy(), 4242, 17, 9, "f64", 0, flow_kind=kind)
        else:
            acc_ref = oracle.pcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, tgt_o, tgt_o, ws, bs, flow.loc.numpy(),
                                           flow.scale.numpy(), 4242, 17, 9, "f64", 0, flow_kind=kind)
        margins = mg.copy()
    got = xd.cpu().numpy()
    moved_g = np.any(np.abs(got - x) > 1e-9 * (1 + np.abs(x)), axis=1)
    moved_r = np.any(xr != x, axis=1)
    assert moved_r[special].all() and moved_g[special].all()
    differ = moved_g != moved_r
    assert differ.sum() <= 8 and np.all(np.abs(margins[differ]) <= 1e-4) and abs(int(n_acc[0]) - acc_ref) <= 8
    v_ll, v_lq = lld.cpu().numpy(), lqd.cpu().numpy()
    assert np.isfinite(v_ll[special]).all() and np.isfinite(v_lq[special]).all() and np.isfinite(lpd.cpu().numpy()[special]).all()
    np.testing.assert_allclose(v_ll[moved_g], tgt_o.logpdf(got)[moved_g], rtol=1e-10, atol=1e-9)
