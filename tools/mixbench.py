import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
n, d = 1_000_000, int(os.environ.get("D", 32))
eng = HipEngine(0, n_max=n, d_max=32)
x = torch.randn((n, d), device="cuda", dtype=torch.float64)
one = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
two = eng.make_mixture([np.log(0.5), np.log(0.5)], np.stack([np.full(d, 0.5), np.full(d, -0.5)]), np.ones((2, d)))
mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
for name, tll in (("C=1", one), ("C=2", two)):
    ll, lp, lq = eng.mixture_logpdf(x, tll), eng.mixture_logpdf(x, one), eng.mixture_logpdf(x, one)
    xx = x.clone()
    eng.pcn_mutate(xx, ll, lp, lq, 0.5, mu, eye, eye, tll, one, one, 7, 0, 0.3, 8, 0, 0.234, True, "f32")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.pcn_mutate(xx, ll, lp, lq, 0.5, mu, eye, eye, tll, one, one, 7, 0, 0.3, 32, 8, 0.234, True, "f32")
    torch.cuda.synchronize(); print(name, "d=%d" % d, round((time.perf_counter() - t0) / 32 * 1e3, 4), "ms/step")
