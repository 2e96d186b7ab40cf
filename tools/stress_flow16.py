"""Stress of the one-kernel flow-proposal steps changed in round 6 for rare faults (a missed hazard behind inline assembly, a race on
the weight stream): many calls of asmc_pcn_mutate_flow on fresh copies of one batch, each followed by the invariants such a fault
would break - every run returns the bits of the first, and the carried log q equals the density kernel's at the returned rows.
Shapes: the flow16 step at D = 64 / 128 (coupling, autoregressive; hidden 64, and 32 / 128 once) and the fused d = 32 step with an
autoregressive proposal; other kernels (a Gram pass, the reference factorisation) run in between.  ITER, N env."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_coupling_flow, random_maf_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402

iters, n = int(os.environ.get("ITER", 150)), int(os.environ.get("N", 40000))
eng = HipEngine(0, n_max=1 << 20, d_max=128)
bad = 0
shapes = [("coupling", 64, 64, 0.0), ("maf", 64, 64, 5.0), ("coupling", 128, 64, 0.0), ("maf", 128, 64, 0.0), ("maf", 32, 64, 0.0),
          ("maf", 32, 64, 4.0), ("coupling", 64, 128, 0.0), ("maf", 128, 32, 0.0), ("maf", 32, 128, 0.0)]
for kind, d, hidden, nu in shapes:
    flow = random_coupling_flow(d, 3, hidden, seed=4) if kind == "coupling" else random_maf_flow(d, 3, hidden, seed=4)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(3)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    t = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(0.05 * np.arange(d) / d)
    A = np.eye(d) + 0.05 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    L, Li = eng.asarray(A), eng.asarray(np.linalg.inv(A))
    first, worst = None, 0.0
    its = iters if hidden == 64 else max(10, iters // 5)
    for it in range(its):
        x = x0.clone()
        ll, lp, lq = eng.mixture_logpdf(x, t), eng.mixture_logpdf(x, t), eng.coupling_logprob(x, dev)
        acc, _, _ = eng.pcn_mutate_flow(x, ll, lp, lq, 0.4, mu, L, Li, t, t, dev, 11, 0, 0.2, 4, 3, 0.234, False, "f64", nu)
        chk = float((lq - eng.coupling_logprob(x, dev)).abs().max())
        worst = max(worst, chk)
        sig = (x.clone(), lq.clone(), acc.copy())
        if first is None:
            first = sig
        elif not (torch.equal(sig[0], first[0]) and torch.equal(sig[1], first[1]) and np.array_equal(sig[2], first[2])):
            bad += 1
            print(f"  MISMATCH {kind} d={d} W={hidden} nu={nu} iteration {it}")
        if chk > 5e-3:
            bad += 1
            print(f"  CARRIED LOG Q OFF {kind} d={d} W={hidden} nu={nu} iteration {it}: {chk}")
        if it % 7 == 0:  # other kernels in between
            eng.mean_gram(x0, n)
    print(f"{kind:8s} d={d:3d} W={hidden:3d} nu={nu}: {its} runs, accept {np.mean(acc) / n:.3f}, max |carried lq - kernel lq| = {worst:.2e}")
print("faults:", bad)
sys.exit(1 if bad else 0)
