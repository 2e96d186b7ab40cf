"""Micro-driver of the flow-proposal mutation step (asmc_pcn_mutate_flow) at 1M x 32: timing per step and the per-kernel
HIP-event table; used under rocprofv3 --pmc by tools/pmc_flowstep.sh.  NOISE=f64|f32, STEPS, N, RHO, ADAPT, KIND=coupling|maf env."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_coupling_flow, random_maf_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import GaussianFlow  # noqa: E402
from aspire_amd.targets import DiagGaussianMixture  # noqa: E402


def main():
    n, d = int(os.environ.get("N", 1_000_000)), 32
    steps, noise = int(os.environ.get("STEPS", 16)), os.environ.get("NOISE", "f64")
    rho0, adapt = float(os.environ.get("RHO", 0.3)), os.environ.get("ADAPT", "1") == "1"  # RHO=0.02 ADAPT=0: ~98 % acceptance
    eng = HipEngine(0, n_max=n, d_max=32)
    w = int(os.environ.get("W", 64))
    flow = random_maf_flow(d, 3, w) if os.environ.get("KIND", "coupling") == "maf" else random_coupling_flow(d, 4, w)
    dev = flow.device_coupling(eng)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    g = GaussianFlow(d, sigma=0.8, seed=3, engine=eng)
    x, _ = g.sample_and_log_prob(n)
    t = lik.device_mixture(eng)
    ll = eng.mixture_logpdf(x, t)
    lp = ll.clone()
    lq = eng.coupling_logprob(x, dev)
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(0.8 * np.eye(d))
    inv = eng.asarray(np.eye(d) / 0.8)
    eng.pcn_mutate_flow(x, ll, lp, lq, 0.5, mu, eye, inv, t, t, dev, 7, 0, rho0, 2, 0, 0.234, adapt, noise)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    acc, _, rho = eng.pcn_mutate_flow(x, ll, lp, lq, 0.5, mu, eye, inv, t, t, dev, 7, 0, rho0, steps, 2, 0.234, adapt, noise)
    e1.record()
    torch.cuda.synchronize()
    print(f"noise={noise} n={n}: {e0.elapsed_time(e1) / steps:.4f} ms/step  accept {acc.mean() / n:.3f} rho {rho:.3f}")
    eng.profile(True)
    eng.pcn_mutate_flow(x, ll, lp, lq, 0.5, mu, eye, inv, t, t, dev, 7, 0, rho0, 4, 40, 0.234, adapt, noise)
    for k, (c, ms) in eng.profile_report().items():
        print(f"   {k:28s} {c:4d} x {ms * 1e3:9.2f} us")


main()
