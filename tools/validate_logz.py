#!/usr/bin/env python
"""Log-evidence check over several seeds (north-star: within 1 sigma of the reference value): for every mutation kernel
family the z-scores (log Z - closed form) / reported error should look standard normal."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import CouplingFlow, GaussianFlow  # noqa: E402
from aspire_amd.samplers.smc import HipSMC  # noqa: E402
from aspire_amd.targets import DiagGaussianMixture  # noqa: E402

n, seeds = int(os.environ.get("N", 1_000_000)), int(os.environ.get("SEEDS", 6))
eng = HipEngine(0, n_max=n, d_max=128)


def closed_form_config5(d):
    # Z = int N(x; 0, I) [0.5 N(x; 2, 0.5 I) + 0.5 N(x; -2, I)] dx = 0.5 N(2 1; 0, 1.5 I) + 0.5 N(-2 1; 0, 2 I)
    a = -0.5 * d * np.log(2 * np.pi * 1.5) - 0.5 * 4 * d / 1.5
    b = -0.5 * d * np.log(2 * np.pi * 2.0) - 0.5 * 4 * d / 2.0
    return np.logaddexp(a, b) + np.log(0.5)


only = [c for c in os.environ.get("CASES", "").split(",") if c]


def case(name, d, lik, prior, flow_fn, true, xp=np, **kw):
    if only and not any(c in name for c in only):
        return
    z, walls = [], []
    for s in range(seeds):
        sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=flow_fn(s), xp=xp, engine=eng,
                    rng=np.random.default_rng(100 + s))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sp.sample(n, sampler_kwargs=dict(n_steps=32, noise=os.environ.get("NOISE", "f64"), **kw), store_sample_history=False)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        z.append((float(out.log_evidence) - true) / float(out.log_evidence_error))
    z = np.array(z)
    print(f"{name:28s} d={d:3d} seeds={seeds} z-scores {np.round(z, 2).tolist()}  mean {z.mean():+.2f}  rms {np.sqrt((z**2).mean()):.2f}"
          f"  wall {np.median(walls):.3f} s")


d = 32
lik = DiagGaussianMixture.isotropic(d, normalized=False)
true32 = 0.5 * d * np.log(np.pi)
gauss = lambda s: GaussianFlow(d, sigma=1.5, seed=s, engine=eng)  # noqa: E731
case("config 2/3 shape, pcn", d, lik, lik, gauss, true32, step_fn="pcn")
case("config 2/3 shape, tpcn", d, lik, lik, gauss, true32)

# arbitrary callables (torch namespace): the split path, whitened-state session for d = 32
tlik = lambda smp: -0.5 * (smp.x * smp.x).sum(1)  # noqa: E731
case("callables, pcn", d, tlik, tlik, gauss, true32, xp=torch, step_fn="pcn")
case("callables, tpcn", d, tlik, tlik, gauss, true32, xp=torch)


def trained_flow(s):
    f = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234 + s)
    f.fit(1.5 * 0.9 * np.random.default_rng(3 + s).normal(size=(8000, d)), n_epochs=8)
    return f


case("config 3, coupling flow + pcn", d, lik, lik, trained_flow, true32, step_fn="pcn")


def trained_maf(s):  # the reference's default flow class
    from aspire_amd.flows import MAFFlow

    f = MAFFlow(d, n_transforms=3, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234 + s)
    f.fit(1.5 * 0.9 * np.random.default_rng(3 + s).normal(size=(8000, d)), n_epochs=8)
    return f


case("config 3, MAF flow + pcn", d, lik, lik, trained_maf, true32, step_fn="pcn")
case("reference defaults: MAF + tpcn", d, lik, lik, trained_maf, true32)  # (round 6: the autoregressive instantiation of the fused step changed)

d6 = 64  # the flow16 step (changed in round 6) inside whole runs: autoregressive proposal at d = 64, tpCN


def trained_maf64(s):
    from aspire_amd.flows import MAFFlow

    f = MAFFlow(d6, n_transforms=3, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234 + s)
    f.fit(1.5 * 0.9 * np.random.default_rng(3 + s).normal(size=(8000, d6)), n_epochs=8)
    return f


lik6 = DiagGaussianMixture.isotropic(d6, normalized=False)
case("d = 64, MAF flow + tpcn (flow16)", d6, lik6, lik6, trained_maf64, 0.5 * d6 * np.log(np.pi))
d5 = 128
lik5 = DiagGaussianMixture(np.stack([2 * np.ones(d5), -2 * np.ones(d5)]), np.stack([0.5 * np.ones(d5), np.ones(d5)]))
prior5 = DiagGaussianMixture.isotropic(d5, 0.0, 1.0)
g5 = lambda s: GaussianFlow(d5, sigma=3.0, engine=eng, seed=4 + s)  # noqa: E731
sub = {"tpcn_fit_subsample": int(os.environ["TPCN_SUB"])} if "TPCN_SUB" in os.environ else {}
case("config 5 (1 GPU), tpcn", d5, lik5, prior5, g5, closed_form_config5(d5), **sub)
case("config 5 (1 GPU), pcn", d5, lik5, prior5, g5, closed_form_config5(d5), step_fn="pcn")
