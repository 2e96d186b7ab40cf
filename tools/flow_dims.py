"""Flow-proposal mutation step by dimension (1M particles, 4 coupling layers of width 64, built-in single-Gaussian targets):
wall time per step of asmc_pcn_mutate_flow and the kernels behind it - d = 32 runs the one-kernel step, every other d the
propose / flow / accept kernels."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_coupling_flow
from aspire_amd.engine import HipEngine
n, steps = int(os.environ.get("N", 1_000_000)), 8
eng = HipEngine(0, n_max=n, d_max=64)
for d in [int(v) for v in os.environ.get("DIMS", "8,16,20,32,48,64").split(",")]:
    flow = random_coupling_flow(d, 4, 64)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(d)
    x = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    t_ll = eng.make_mixture([0.3], np.full((1, d), 0.25), np.ones((1, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(np.zeros(d)); L = eng.asarray(np.eye(d)); Linv = eng.asarray(np.eye(d))
    ll, lp, lq = eng.mixture_logpdf(x, t_ll), eng.mixture_logpdf(x, t_lp), eng.coupling_logprob(x, dev)
    def run():
        return eng.pcn_mutate_flow(x, ll, lp, lq, 0.35, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, 0.2, steps, 5, 0.234, False, "f64", 0.0)
    run(); torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rep = eng.profile_report(); eng.profile(False)
    top = sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:4]
    print(f"d={d:3d}: {dt / steps * 1e3:7.3f} ms/step  " + "  ".join(f"{k}={c}x{ms * 1e3:.0f}us" for k, (c, ms) in top))
