"""Micro-driver of the fused importance step (asmc_importance_step + gather) at N x 32: wall time per step and the
per-kernel HIP-event table.  N, STEPS env."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from aspire_amd import smc_math  # noqa: E402
from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import GaussianFlow  # noqa: E402
from aspire_amd.targets import DiagGaussianMixture  # noqa: E402


def main():
    n, d = int(os.environ.get("N", 1 << 20)), 32
    steps = int(os.environ.get("STEPS", 40))
    eng = HipEngine(0, n_max=n, d_max=32)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    g = GaussianFlow(d, sigma=1.5, seed=0, engine=eng)
    x, lq = g.sample_and_log_prob(n)
    ll = eng.mixture_logpdf(x, lik.device_mixture(eng))
    lp = ll.clone()
    rng = np.random.default_rng(12345)

    def step():
        idx = eng.importance_step(ll, lp, lq, 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
        rows = eng.gather(idx, x, ll, lp, lq)
        res = eng.importance_result()
        rng.bit_generator.advance(n)
        return rows, res

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rows, res = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"importance step: {dt * 1e3:.4f} ms  beta*={res[0]} rounds={res[3]}")
    eng.profile(True)
    for _ in range(10):
        step()
    rep = eng.profile_report()
    eng.profile(False)
    for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        print(f"  {k:28s} {c / 10:5.1f} x {ms * 1e3:8.2f} us")


if __name__ == "__main__":
    main()
