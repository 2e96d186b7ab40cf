"""Log-evidence validation of every mutation kernel family, in the GPU suite.

SURVEY.md §8 a12: the pCN / tpCN kernels follow this repository's own specification (minipcn is absent from the reference tree:
parity unpinned), so their only EXTERNAL check is the quantity the sampler exists to compute - log Z against its closed form,
in units of the sampler's own error estimate (north star: "log-evidence within 1 sigma of reference").  `tools/validate_logz.py`
prints 30 seeds per family at 1M particles; this is its reduced form for every driver run: 16 seeds per family at 250k particles,
z = (log Z - closed form) / log_evidence_error must look standard normal: |mean z| < 0.6 (2.4 standard errors of the mean) and
rms z in [0.55, 1.5] (a chi-square with 16 degrees of freedom leaves that band with probability 2 %).
Runs are bit-reproducible for a fixed seed (counter-based noise, seeded numpy generator), so a family either passes or fails - it
does not flake.  Reference semantics: `/root/reference/src/aspire/samplers/smc/base.py:400-488` (evidence accumulation),
`/root/reference/src/aspire/samplers/smc/minipcn.py:69-135` (mutation).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, SEEDS = 250_000, 16


@pytest.fixture(scope="module")
def eng():
    from aspire_amd.engine import HipEngine

    return HipEngine(0, n_max=N, d_max=128)


def closed_form_config5(d):
    # Z = int N(x; 0, I) [0.5 N(x; 2 1, 0.5 I) + 0.5 N(x; -2 1, I)] dx = 0.5 N(2 1; 0, 1.5 I) + 0.5 N(-2 1; 0, 2 I)
    a = -0.5 * d * np.log(2 * np.pi * 1.5) - 0.5 * 4 * d / 1.5
    b = -0.5 * d * np.log(2 * np.pi * 2.0) - 0.5 * 4 * d / 2.0
    return float(np.logaddexp(a, b) + np.log(0.5))


def z_scores(eng, d, lik, prior, flow_fn, true, xp=np, n=N, seeds=SEEDS, **kw):
    from aspire_amd.samplers.smc import HipSMC

    z, paths = [], set()
    for s in range(seeds):
        sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=flow_fn(s), xp=xp, engine=eng,
                    rng=np.random.default_rng(100 + s))
        out = sp.sample(n, sampler_kwargs=dict(n_steps=32, **kw), store_sample_history=False)
        z.append((float(out.log_evidence) - true) / float(out.log_evidence_error))
        paths.add(sp.last_mutation_path)
    return np.array(z), paths


def check(z, what):
    mean, rms = float(z.mean()), float(np.sqrt((z**2).mean()))
    assert abs(mean) < 0.6 and 0.55 <= rms <= 1.5, f"{what}: z = {np.round(z, 2).tolist()} mean {mean:+.2f} rms {rms:.2f}"


@pytest.fixture(scope="module")
def gauss32(eng):
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.targets import DiagGaussianMixture

    d = 32
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    return d, lik, (lambda s: GaussianFlow(d, sigma=1.5, seed=s, engine=eng)), 0.5 * d * math.log(math.pi)


@pytest.mark.parametrize("step_fn", ["pcn", "tpcn"])
def test_logz_builtin_densities(eng, gauss32, step_fn):
    """configs[1]/[2] shape, analytic proposal: the register-resident pCN / tpCN kernels (asmc_pcn_mutate)."""
    d, lik, flow_fn, true = gauss32
    z, paths = z_scores(eng, d, lik, lik, flow_fn, true, step_fn=step_fn)
    assert all("built-in densities" in p for p in paths), paths
    check(z, f"built-in densities, {step_fn}")


@pytest.mark.parametrize("step_fn", ["pcn", "tpcn"])
def test_logz_python_callables(eng, gauss32, step_fn):
    """Arbitrary Python densities (torch namespace): the split propose / accept path on the whitened state."""
    d, _, flow_fn, true = gauss32
    tlik = lambda smp: -0.5 * (smp.x * smp.x).sum(1)  # noqa: E731
    z, paths = z_scores(eng, d, tlik, tlik, flow_fn, true, xp=torch, step_fn=step_fn)
    assert all("callables" in p for p in paths), paths
    check(z, f"callables, {step_fn}")


@pytest.mark.parametrize("flow_cls", ["coupling", "maf"])
def test_logz_flow_proposal_fused_step(eng, gauss32, flow_cls):
    """configs[2]: trained neural proposal, the one-kernel flow-proposal step (k_pcn_flow_fused), coupling layers and - the
    reference's default flow class - masked autoregressive transforms."""
    from aspire_amd.flows import CouplingFlow, MAFFlow

    d, lik, _, true = gauss32
    if flow_cls == "coupling":
        flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
    else:
        flow = MAFFlow(d, n_transforms=3, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
    flow.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)

    def flow_fn(s):  # one trained flow, a fresh draw stream per run
        flow._hip_draws = 1000 * s
        return flow

    eng.profile(True)
    z, paths = z_scores(eng, d, lik, lik, flow_fn, true, step_fn="pcn")
    rep = eng.profile_report()
    eng.profile(False)
    assert all("flow: device-side step loop" in p for p in paths), paths
    assert "k_pcn_flow_fused" in rep and not any(k.startswith(("k_coupling_logprob", "k_maf_logprob")) for k in rep), sorted(rep)
    check(z, f"{flow_cls} flow proposal, fused step")


@pytest.mark.parametrize("step_fn", ["tpcn", "pcn"])
def test_logz_config5_mixture_d128(eng, step_fn):
    """BASELINE configs[4] on one GPU: d = 128, two-component Gaussian-mixture likelihood (mu = +-2, cov I/2 and I), N(0, I)
    prior, q = N(0, 9 I), adaptive tempering; the fp64 matrix-core kernels (k_pcn_mm)."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.targets import DiagGaussianMixture

    d = 128
    lik = DiagGaussianMixture(np.stack([2 * np.ones(d), -2 * np.ones(d)]), np.stack([0.5 * np.ones(d), np.ones(d)]))
    prior = DiagGaussianMixture.isotropic(d, 0.0, 1.0)
    z, _ = z_scores(eng, d, lik, prior, lambda s: GaussianFlow(d, sigma=3.0, engine=eng, seed=4 + s), closed_form_config5(d),
                    n=N, step_fn=step_fn)
    check(z, f"config 5 (d = 128 mixture), {step_fn}")
