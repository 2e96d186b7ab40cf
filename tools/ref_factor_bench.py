import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aspire_amd.engine import HipEngine
eng = HipEngine(0, n_max=1 << 18, d_max=128)
for d in (32, 64, 128):
    n = 20000
    x = torch.randn((n, d), device=eng.device, dtype=torch.float64)
    s, g = eng.mean_gram(x, n)
    for _ in range(3): eng.reference_factor(d, n, n, moments=(s, g))
    eng.profile(True)
    for _ in range(10): eng.reference_factor(d, n, n, moments=(s, g))
    rep = eng.profile_report(); eng.profile(False)
    print(os.environ.get("ASMC_REF_THREADS", "256"), d, round(rep["k_ref_factor"][1] * 1e3, 1), "us")
