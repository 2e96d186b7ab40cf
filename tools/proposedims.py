import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
n = 1_000_000
eng = HipEngine(0, n_max=n, d_max=32)
for d in (6, 10, 16, 20, 24, 31, 32):
    x = torch.randn((n, d), device="cuda", dtype=torch.float64)
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
    for _ in range(3): eng.pcn_propose(x, mu, eye, eye, 0.3, 5, 0, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(20): eng.pcn_propose(x, mu, eye, eye, 0.3, 5, 0, t)
    torch.cuda.synchronize(); print(d, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms")
