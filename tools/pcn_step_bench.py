"""Micro-driver of the register-resident pCN / tpCN step (k_pcn_reg_y / k_tpcn_reg_y: built-in Gaussian targets, no flow) at 1M x D,
D = 32 (or 4 / 8 / 16): NU=0|5, NOISE=f64|f32, STEPS; prints the per-kernel HIP-event table.  Same-box A/B through ASMC_LIB_PATH."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aspire_amd.engine import HipEngine  # noqa: E402

n, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 32))
nu, steps, noise = float(os.environ.get("NU", 0.0)), int(os.environ.get("STEPS", 32)), os.environ.get("NOISE", "f64")
eng = HipEngine(0, n_max=n, d_max=max(d, 32))
g = torch.Generator(eng.device).manual_seed(d)
x = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
t = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
tq = eng.make_mixture([0.0], np.zeros((1, d)), 0.5 * np.ones((1, d)))
rng = np.random.default_rng(3)
A = rng.normal(size=(d, d)) * 0.1 + np.eye(d)
L = np.linalg.cholesky(A @ A.T)
mu, Ld, Li = eng.asarray(np.zeros(d)), eng.asarray(L), eng.asarray(np.linalg.inv(L))
ll, lp, lq = eng.mixture_logpdf(x, t), eng.mixture_logpdf(x, t), eng.mixture_logpdf(x, tq)
args = (x, ll, lp, lq, 0.35, mu, Ld, Li, t, t, tq, 77, 1000, 0.2, steps, 5, 0.234, False, noise, nu)
eng.pcn_mutate(*args)
torch.cuda.synchronize()
eng.profile(True)
acc, _, _ = eng.pcn_mutate(*args)
for k, (c, ms) in sorted(eng.profile_report().items(), key=lambda kv: -kv[1][0] * kv[1][1])[:4]:
    print(f"   {k:24s} {c:4d} x {ms * 1e3:9.2f} us")
print(f"d={d} nu={nu} noise={noise}: accept {np.mean(acc) / n:.3f}")
