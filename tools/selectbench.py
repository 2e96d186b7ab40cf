"""k_pcg64_select timing vs the number of draws (what every rank walks in owner-layout resampling)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
from aspire_amd.smc_math import pcg64_state
eng = HipEngine(0, n_max=1 << 20, d_max=32)
st = pcg64_state(np.random.default_rng(1))
for n_total, world in ((1 << 20, 1), (2_000_000, 2), (4_000_000, 4), (8_000_000, 8)):
    lo, hi = 0.25 / world, 1.25 / world if world > 1 else 1.0
    for _ in range(3):
        eng.pcg64_select(st, n_total, lo, min(hi, 1.0))
    eng.profile(True)
    for _ in range(10):
        q = eng.pcg64_select(st, n_total, lo, min(hi, 1.0))
    rep = eng.profile_report(); eng.profile(False)
    print(n_total, world, q.numel(), {k: round(v[1] * 1e3, 1) for k, v in rep.items()})
