import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from aspire_amd import smc_math
from aspire_amd.engine import HipEngine
from aspire_amd.flows import GaussianFlow
from aspire_amd.targets import DiagGaussianMixture
n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
g = GaussianFlow(d, sigma=1.5, seed=0, engine=eng)
x, lq = g.sample_and_log_prob(n)
ll = eng.mixture_logpdf(x, lik.device_mixture(eng)); lp = ll.clone()
rng = np.random.default_rng(1)
for _ in range(4):
    idx = eng.importance_step(ll, lp, lq, 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng), n)
    print(eng.importance_result()[:4])
