import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
g = np.random.default_rng(0)
x0 = torch.as_tensor(g.normal(size=(n, d)) * np.sqrt(0.5), device="cuda")
tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
def run(name, L, rho, seed, beta=1.0, adapt=True):
    x = x0.clone(); ll = eng.mixture_logpdf(x, tgt); lp = ll.clone(); lq = eng.mixture_logpdf(x, q)
    Ld, Li, mu = eng.asarray(np.tril(L)), eng.asarray(np.tril(np.linalg.inv(L))), eng.asarray(np.zeros(d))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    na, rh, r = eng.pcn_mutate(x, ll, lp, lq, beta, mu, Ld, Li, tgt, tgt, q, seed, 0, rho, 32, 0, 0.234, adapt, "f32")
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter()-t0)*1e3/32*1e3:8.1f} us/step  acc {na.mean()/n:.3f} rho_end {r:.3f}")
A = g.normal(size=(d, d)); cov = A @ A.T / d + 0.5 * np.eye(d)
for rep in range(2):
    run("identity rho.3 seed1", np.eye(d), 0.3, 1)
    run("identity rho.42 bigseed", np.eye(d), 0.42, 2**62 + 12345)
    run("sqrt.5*I (exact ref)", np.sqrt(0.5) * np.eye(d), 0.42, 7)
    run("dense chol", np.linalg.cholesky(cov), 0.42, 7)
    run("sqrt.5*I noadapt rho.99", np.sqrt(0.5) * np.eye(d), 0.99, 7, adapt=False)
    run("sqrt.5*I beta.1", np.sqrt(0.5) * np.eye(d), 0.42, 7, beta=0.1)
