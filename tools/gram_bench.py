"""Gram kernel at 1M x 32 fp64 / fp32: time per launch, and the result against the LDS-tile kernel's (ASMC_GRAM_LDS32=1 in a second process)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
eng = HipEngine(0, n_max=1 << 20, d_max=64)
D = int(os.environ.get("D", 32))
g = torch.Generator(eng.device).manual_seed(1)
for dt in (torch.float64, torch.float32):
    x = (0.3 + torch.randn((1_000_003, D), device=eng.device, dtype=torch.float64, generator=g)).to(dt)[:1_000_001].contiguous()
    n = x.shape[0]
    for _ in range(5): s, gr = eng.mean_gram(x, n)
    eng.profile(True)
    for _ in range(20): eng.mean_gram(x, n)
    rep = eng.profile_report(); eng.profile(False)
    print(D, "lds" if os.environ.get("ASMC_GRAM_LDS32") else "stream", dt, "k_gram_mm", round(rep["k_gram_mm"][1] * 1e3, 1), "us  checksum", repr(float(np.abs(gr).sum())), repr(float(gr[17, 3])), repr(float(gr[3, 17])))
