"""The reference's likelihood-hole integration test on the CPU test double of the engine.

/root/reference/tests/integration_tests/test_integration.py:131-166 (`test_smc_log_likelihood_with_invalid_value`): a 2-D
Gaussian likelihood (mean 2, unit variance; conftest.py:34-97), a uniform prior on [-10, 10]^2 (conftest.py:100-109), the
likelihood replaced by -inf, NaN or +inf inside r < 1, then `Aspire.fit` and `sample_posterior(sampler="smc")` with the
default step (tpCN).  The reference's CI expects all three to finish.  -inf and NaN are rejected by the reference's own
NaN -> -inf rule (smc/base.py:518); +inf must be rejected inside the third-party step (minipcn, absent here), which this
repository specifies itself: a proposal whose tempered log-target is +inf is rejected and counted as a rejection
(csrc/asmc_pcn_dev.h `log_p_t`, oracle/asmc_oracle.c `orc_log_p_t`) - unverified against the package.

Here: the same construction through `aspire_amd.Aspire` with Python callables (the split path) on the oracle-backed test
double; tests/test_gpu_hole.py runs it on the HIP engine, callables and built-in targets.
"""
import math

import numpy as np
import pytest

from oracle_engine import OracleEngine

HOLE_VALUES = [float("-inf"), float("nan"), float("inf")]
IDS = ["minus_inf", "nan", "plus_inf"]


def hole_problem(value, d=2, mean=2.0, std=1.0, lo=-10.0, hi=10.0, radius=1.0):
    """The reference fixtures (conftest.py:34-109) + the hole of test_integration.py:141-146, numpy callables."""

    def log_likelihood(samples):
        x = np.asarray(samples.x, dtype=np.float64)
        assert x.shape[-1] == d
        const = math.log(1.0 / (std * math.sqrt(2 * math.pi)))
        return np.sum(const - 0.5 * ((x - mean) / std) ** 2, axis=-1)

    def log_prior(samples):
        x = np.asarray(samples.x, dtype=np.float64)
        val = np.where((x >= lo) & (x <= hi), math.log(1.0 / (hi - lo)), -np.inf)
        return np.sum(val, axis=-1)

    def log_likelihood_with_hole(samples):
        logl = log_likelihood(samples)
        r = np.linalg.norm(np.asarray(samples.x, dtype=np.float64), axis=1)
        return np.where(r < radius, value, logl)

    return log_likelihood_with_hole, log_prior


def log_z_without_hole(d=2, mean=2.0, lo=-10.0, hi=10.0, radius=1.0):
    """log of  int_{box, r >= radius} N(x; mean, I) / (hi - lo)^d dx:  the mass of the box is 1 to 1e-15, the mass of the disc
    is the non-central chi-square cdf with d degrees of freedom and non-centrality d mean^2 at radius^2."""
    from scipy.stats import ncx2

    return float(np.log1p(-ncx2.cdf(radius**2, d, d * mean**2)) - d * math.log(hi - lo))


def run_hole(value, engine, n_samples, seed, device=None, xp=np, **sample_kwargs):
    from aspire_amd import Aspire, Samples

    d = 2
    log_like, log_prior = hole_problem(value, d)
    params = [f"x_{i}" for i in range(d)]
    g = np.random.default_rng(seed)
    train = Samples(g.normal(2.0, 1.0, size=(500, d)), parameters=params, xp=np)
    with np.errstate(all="ignore"):
        assert not np.isfinite(log_like(train)).all()  # the reference's own precondition (test_integration.py:160)
    asp = Aspire(log_likelihood=log_like, log_prior=log_prior, dims=d, parameters=params,
                 prior_bounds={p: [-10, 10] for p in params}, bounded_to_unbounded=False, flow_backend="gaussian", xp=xp,
                 engine=engine, seed=seed + 1, device=device)
    asp.fit(train)
    out, history = asp.sample_posterior(n_samples=n_samples, sampler="smc", return_history=True, engine=engine,
                                        rng=np.random.default_rng(seed + 2), **sample_kwargs)
    return asp, out, history


def check_hole_run(asp, out, history, n_samples, sigmas=3.0, slack=0.05):
    assert history.beta[-1] == 1.0
    ll = np.asarray(out.log_likelihood, dtype=np.float64)
    x = np.asarray(out.x, dtype=np.float64)
    assert len(x) == n_samples and np.isfinite(ll).all() and np.isfinite(np.asarray(out.log_prior)).all()
    assert (np.linalg.norm(x, axis=1) >= 1.0).all()  # nobody settled in the hole
    assert np.isfinite(out.log_evidence) and np.isfinite(out.log_evidence_error)
    # (the initial draw drops the proposal's rows inside the hole, mcmc.py:88-90: the estimator then carries the proposal's mass
    # of the disc, about one per cent of Z - the slack)
    assert abs(float(out.log_evidence) - log_z_without_hole()) < sigmas * float(out.log_evidence_error) + slack, (
        float(out.log_evidence), log_z_without_hole(), float(out.log_evidence_error))
    assert 0.02 < np.mean(history.mcmc_acceptance) < 0.98


@pytest.mark.parametrize("value", HOLE_VALUES, ids=IDS)
def test_smc_log_likelihood_with_invalid_value_on_the_test_double(value):
    n = 300
    asp, out, history = run_hole(value, OracleEngine(), n, seed=11)
    check_hole_run(asp, out, history, n)
    assert asp.sampler.last_mutation_path is not None


@pytest.mark.parametrize("value", HOLE_VALUES, ids=IDS)
def test_log_p_t_maps_nan_and_plus_inf_to_minus_inf(value):
    """The rule itself, oracle and host: (1 - beta) log q + beta (ll + lp), NaN and +inf -> -inf (smc/base.py:518 + this
    repository's reading of the third-party step)."""
    import oracle as O

    assert O.log_p_t(value, -1.0, -2.0, 0.3) == -np.inf
    assert O.log_p_t(-1.0, -1.0, value, 0.3) == -np.inf
    assert O.log_p_t(-1.0, -1.5, -2.0, 0.25) == (1 - 0.25) * -2.0 + 0.25 * (-1.0 + -1.5)
