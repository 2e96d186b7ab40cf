#!/usr/bin/env python
"""BASELINE config 5 on one GPU: N x 128, two-component Gaussian-mixture likelihood (mu = +-2, cov 1/2 I and I),
N(0, I) prior, q = N(0, 3^2 I), adaptive tempering, pCN on the fp64 matrix cores.  Prints wall time, throughput, the
log-evidence against its closed form and the per-kernel device times."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import GaussianFlow  # noqa: E402
from aspire_amd.samplers.smc import HipSMC  # noqa: E402
from aspire_amd.targets import DiagGaussianMixture  # noqa: E402

n, d, steps = int(os.environ.get("N", 1_000_000)), 128, int(os.environ.get("STEPS", 32))
eng = HipEngine(0, n_max=n, d_max=d)
lik = DiagGaussianMixture(np.stack([2 * np.ones(d), -2 * np.ones(d)]), np.stack([0.5 * np.ones(d), np.ones(d)]))
prior = DiagGaussianMixture.isotropic(d, 0.0, 1.0)


def run(nn, profile):
    sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=GaussianFlow(d, sigma=3.0, engine=eng, seed=4),
                xp=np, engine=eng, rng=np.random.default_rng(1))
    eng.profile(profile)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sp.sample(nn, sampler_kwargs=dict(n_steps=steps, noise=os.environ.get("NOISE", "f32"), step_fn=os.environ.get("STEP_FN", "tpcn")), store_sample_history=False)
    torch.cuda.synchronize()
    return sp, out, time.perf_counter() - t0


if not os.environ.get("NOWARM"):  # (NOWARM=1 under rocprofv3 --pmc: every launch of the trace is then a 1M-particle launch)
    run(min(n, 65536), False)  # warm
sp, out, dt = run(n, True)
rep = eng.profile_report()
eng.profile(False)


def lg(mu, var):
    return -0.5 * d * np.log(2 * np.pi * var) - 0.5 * d * mu * mu / var


true = np.logaddexp(np.log(0.5) + lg(2.0, 1.5), np.log(0.5) + lg(2.0, 2.0))
nt = len(sp.history.beta)
print(f"N={n} d={d}: wall {dt:.3f} s, {nt} temperatures x {steps} pCN steps = {n*nt*steps/dt/1e9:.3f} G particle-steps/s")
print(f"log Z = {float(out.log_evidence):.4f} +- {float(out.log_evidence_error):.4f}   closed form {true:.4f}   "
      f"mean accept {np.mean(sp.history.mcmc_acceptance):.3f}")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:int(os.environ.get("TOP", 10))]:
    print(f"  {k:24s} n={v[0]:5d} avg_us={v[1]*1e3:9.1f} total_ms={v[0]*v[1]:8.2f}")
