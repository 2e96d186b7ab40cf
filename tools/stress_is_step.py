#!/usr/bin/env python
"""Repeatability stress of the importance step (asmc_importance_step + gather): the same inputs and generator state REPS times;
every repetition must return the very same indices, rows and scalars (a race in the scans / the chain / the guide fill would
show up as a differing hash).  N, REPS env."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd import smc_math  # noqa: E402
from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import GaussianFlow  # noqa: E402
from aspire_amd.targets import DiagGaussianMixture  # noqa: E402

n, reps, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("REPS", 3000)), 32
eng = HipEngine(0, n_max=n, d_max=d)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
x, lq = GaussianFlow(d, sigma=1.5, seed=0, engine=eng).sample_and_log_prob(n)
ll = eng.mixture_logpdf(x, lik.device_mixture(eng))
lp = ll.clone()
state = smc_math.pcg64_state(np.random.default_rng(12345))
w = torch.arange(1, n + 1, device=eng.device, dtype=torch.int64)
first, bad = None, 0
for r in range(reps):
    idx = eng.importance_step(ll, lp, lq, 0.0, 0.5, 1e-6, state, n)
    rows = eng.gather(idx, x, ll, lp, lq)
    res = eng.importance_result()
    h = (int((idx * w).sum().item()), float(rows[0].sum().item()), float(rows[3].sum().item()), res[0], res[3])
    if first is None:
        first = h
    elif h != first:
        bad += 1
        print("MISMATCH at repetition", r, h, first)
print(f"{reps} repetitions at n = {n}: {bad} mismatches; beta* = {first[3]}, rounds = {first[4]}")
sys.exit(1 if bad else 0)
