"""Stress of the fused flow-proposal step for rare faults: many calls of asmc_pcn_mutate_flow on fresh copies of one batch,
each followed by the invariants a race or a missed hazard would break - the carried log q / log-likelihood equal the densities
recomputed at the returned positions, and every run returns the same bits as the first.  Other kernels (Gram on the matrix
cores, the reference factorisation with its 132 KB of LDS) run in between.  ITER, N, HIDDEN env."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_coupling_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402

iters = int(os.environ.get("ITER", 300))
eng = HipEngine(0, n_max=1 << 20, d_max=128)
d, n_steps, beta, rho = 32, 5, 0.35, 0.4
bad = 0
for hidden, n in [(int(h), int(m)) for h in os.environ.get("HIDDEN", "128,64").split(",") for m in os.environ.get("N", "640,100000").split(",")]:
    flow = random_coupling_flow(d, 4 if hidden < 128 else 1, hidden)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(3)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    t_ll = eng.make_mixture([0.3], np.full((1, d), 0.25), np.ones((1, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(0.1 * np.arange(d) / d)
    A = np.eye(d) + 0.05 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    L, Linv = eng.asarray(A), eng.asarray(np.linalg.inv(A))
    big = torch.randn((70000, 128), device=eng.device, dtype=torch.float64, generator=g)
    ll0, lp0, lq0 = eng.mixture_logpdf(x0, t_ll), eng.mixture_logpdf(x0, t_lp), eng.coupling_logprob(x0, dev)
    first = None
    for it in range(iters):
        x, ll, lp, lq = x0.clone(), ll0.clone(), lp0.clone(), lq0.clone()
        if it % 3 == 1:
            s, gr = eng.mean_gram(big, 70000)
            eng.reference_factor(128, 70000, 70000, moments=(s, gr))
        n_acc, _, _ = eng.pcn_mutate_flow(x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho, n_steps, 5, 0.234,
                                          False, "f64", 0.0)
        lq_re = eng.coupling_logprob(x, dev)
        ll_re = eng.mixture_logpdf(x, t_ll)
        e_lq = (lq - lq_re).abs().max().item()
        e_ll = (ll - ll_re).abs().max().item()
        cur = (x.clone(), lq.clone(), np.array(n_acc))
        same = first is None or (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]) and np.array_equal(cur[2], first[2]))
        if first is None:
            first = cur
        if e_lq > 2e-3 or e_ll > 1e-8 or not same:
            bad += 1
            nd = int((cur[0] != first[0]).any(dim=1).sum())
            print(f"hidden {hidden} n {n} iteration {it}: |lq - recomputed| {e_lq:.3g}  |ll - recomputed| {e_ll:.3g}  "
                  f"rows differing from the first run {nd}  accepts {cur[2].tolist()} vs {first[2].tolist()}", flush=True)
    print(f"hidden {hidden} n {n}: {iters} runs done, faults so far {bad}", flush=True)
print("FAULTS", bad)
