"""Where k_exact_chain's time goes: the exact cdf of 1M normalised weights (the running sum ends at 1 = a binade edge: the last
tiles fail their predicted-binade check and are scanned element-wise) against the same weights scaled by 0.9 (no edge near the
end) and by 0.45, per kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
n = int(os.environ.get("N", 1_000_000))
eng = HipEngine(0, n_max=n, d_max=32)
g = np.random.default_rng(0)
lw = -0.5 * g.chisquare(32, n) * 0.07
w = np.exp(lw - lw.max()); w /= w.sum()
for scale in (1.0, 0.9, 0.45):
    wd = eng.asarray(w * scale)
    for _ in range(5): eng.cdf(wd, "exact")
    eng.profile(True)
    for _ in range(20): eng.cdf(wd, "exact")
    rep = eng.profile_report(); eng.profile(False)
    print(f"scale {scale}: " + "  ".join(f"{k}={ms*1e3:.1f}us" for k, (c, ms) in sorted(rep.items())))
