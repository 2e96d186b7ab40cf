"""Does the tpCN kernel leave its target invariant at every d?  Start from exact samples of N(0, I/2) (the beta = 1 target of
ll = lp = -|x|^2/2 ... here ll = lp = -0.5 |x|^2 with beta = 1 and q irrelevant), run STEPS tpCN steps, look at the moments.
Env: DIMS (comma list), NU, NOISE, STEPS, N."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aspire_amd.engine import HipEngine

n = int(os.environ.get("N", 400000))
steps = int(os.environ.get("STEPS", 64))
noise = os.environ.get("NOISE", "f32")
nus = [float(v) for v in os.environ.get("NU", "4,30").split(",")]
eng = HipEngine(0, n_max=n, d_max=128)
for d in [int(v) for v in os.environ.get("DIMS", "32,64,128").split(",")]:
    for nu in nus:
        for step_adapt in (True, False):
            g = np.random.default_rng(1)
            x = g.normal(size=(n, d)) * np.sqrt(0.5)
            tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
            q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
            xd = eng.asarray(x)
            ll = eng.mixture_logpdf(xd, tgt); lp = ll.clone(); lq = eng.mixture_logpdf(xd, q)
            L = np.diag(np.full(d, 0.8))
            n_acc, rho_hist, rho = eng.pcn_mutate(xd, ll, lp, lq, 1.0, eng.asarray(np.full(d, 0.1)), eng.asarray(L),
                                                  eng.asarray(np.linalg.inv(L)), tgt, tgt, q, 11, 0, 0.3, steps, 0, 0.234, step_adapt, noise, nu)
            xs = xd.cpu().numpy()
            r2 = (xs ** 2).sum(1)
            # |x|^2 ~ 0.5 chi2_d: mean d/2, var d/2
            se = np.sqrt(0.5 * d / n)
            print(f"d={d:3d} nu={nu:5.1f} adapt={step_adapt!s:5s} acc={np.mean(n_acc)/n:.3f} rho={rho:.3f}  mean|x|^2 - d/2 = {r2.mean() - 0.5*d:+.5f} ({(r2.mean()-0.5*d)/se:+.2f} se)"
                  f"  var|x|^2 / (d/2) = {r2.var()/(0.5*d):.4f}  max|mean_j| = {np.abs(xs.mean(0)).max():.4f}")
