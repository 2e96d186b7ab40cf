import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
x = torch.randn((n, d), device="cuda", dtype=torch.float64)
tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
ll = eng.mixture_logpdf(x, tgt); lp = ll.clone(); lq = ll.clone()
mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
ts = []
t00 = time.perf_counter()
for i in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.pcn_mutate(x, ll, lp, lq, 0.5, mu, eye, eye, tgt, tgt, tgt, 1, 0, 0.3, 32, 0, 0.234, True, "f32")
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    mode = os.environ.get("ALLOC", "")
    if mode == "cache":
        y = torch.empty((n, d), device="cuda", dtype=torch.float64); del y
    elif mode == "fresh":
        t1 = time.perf_counter(); y = torch.empty((n + i, d), device="cuda", dtype=torch.float64); y.zero_(); del y; torch.cuda.empty_cache()
        torch.cuda.synchronize(); ts[-1] = round(ts[-1], 1) + 1000 * round((time.perf_counter() - t1) * 1e3)  # encode alloc ms in thousands
print("total", time.perf_counter() - t00)
print([round(t, 1) for t in ts])
