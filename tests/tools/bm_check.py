"""Accuracy of the kernels' Box-Muller normals and of the oracle's libm version against an 80-bit long-double evaluation of
the same Philox words (who is closer to the exact value, and by how many ulp)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle as O
from aspire_amd.engine import HipEngine
eng = HipEngine(0, n_max=1 << 16, d_max=32)
d, n = 32, 2048
x, _ = eng.gaussian_draw(n, d, torch.float64, eng.asarray(np.zeros(d)), eng.asarray(np.ones(d)), 99, 5, 1, want_lq=False)
xn = x.cpu().numpy()
ref = np.stack([O.pcn_noise(99, 5 + i, 1, d)[0] for i in range(n)])
LD = np.longdouble
exact = np.zeros((n, d), dtype=LD)
two_pi = LD(2) * np.arctan2(LD(0), LD(-1))
for i in range(n):
    for p in range(d // 2):
        w = O.philox4x32_10([(5 + i) & 0xFFFFFFFF, (5 + i) >> 32, 1, p], [99, 0])
        def u01(hi, lo):
            v = ((int(hi) << 21) ^ (int(lo) >> 11)) & ((1 << 53) - 1)
            return np.float64(np.float64(v) + 0.5) * np.float64(1.0 / 9007199254740992.0)
        u1, u2 = LD(u01(w[0], w[1])), LD(u01(w[2], w[3]))
        r = np.sqrt(LD(-2) * np.log(u1))
        exact[i, 2 * p], exact[i, 2 * p + 1] = r * np.cos(two_pi * u2), r * np.sin(two_pi * u2)
ex64 = exact.astype(np.float64)
ulp = np.spacing(np.abs(ex64))
for name, v in (("kernel", xn), ("oracle libm", ref)):
    e = np.abs((v.astype(LD) - exact).astype(np.float64))
    print(f"{name:12s} max abs err {e.max():.3e}  max err in ulp of the value {np.max(e / ulp):.2f}  mean {np.mean(e / ulp):.3f}  frac > 2 ulp {np.mean(e / ulp > 2):.4f}")
