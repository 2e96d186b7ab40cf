"""Flow-proposal mutation step by dimension (1M particles, width-64 flows, built-in single-Gaussian targets): wall time per step of
asmc_pcn_mutate_flow and the kernels behind it - d <= 32: the one-kernel step k_pcn_flow_fused (narrower problems zero-padded);
32 < d <= 128 (round 5): the one-kernel step on 16-particle groups with streamed weights, k_pcn_flow16 (padded to 64 / 128).
KINDS=coupling,maf  DIMS=...  NU=0|5 (pCN / tpCN)  ASMC_FLOW16_OFF=1 gives round 4's multi-kernel path above 32 dimensions."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_coupling_flow, random_maf_flow
from aspire_amd.engine import HipEngine
n, steps = int(os.environ.get("N", 1_000_000)), 8
nu = float(os.environ.get("NU", 0.0))
eng = HipEngine(0, n_max=n, d_max=128)
for kind in os.environ.get("KINDS", "coupling,maf").split(","):
    for d in [int(v) for v in os.environ.get("DIMS", "8,16,20,32,48,64,100,128").split(",")]:
        try:
            flow = random_coupling_flow(d, 4, 64) if kind == "coupling" else random_maf_flow(d, 3, 64)
            dev = flow.device_coupling(eng)
        except Exception as exc:
            print(f"{kind:8s} d={d:3d}: no device path ({exc})")
            continue
        g = torch.Generator(eng.device).manual_seed(d)
        x = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
        t_ll = eng.make_mixture([0.3], np.full((1, d), 0.25), np.ones((1, d)) * 1.5)
        t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
        mu = eng.asarray(np.zeros(d)); L = eng.asarray(np.eye(d)); Linv = eng.asarray(np.eye(d))
        ll, lp, lq = eng.mixture_logpdf(x, t_ll), eng.mixture_logpdf(x, t_lp), eng.coupling_logprob(x, dev)
        def run():
            return eng.pcn_mutate_flow(x, ll, lp, lq, 0.35, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, 0.2, steps, 5, 0.234, False, "f64", nu)
        run(); torch.cuda.synchronize()
        eng.profile(True)
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rep = eng.profile_report(); eng.profile(False)
        top = sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:4]
        print(f"{kind:8s} d={d:3d}: {dt / steps * 1e3:7.3f} ms/step  " + "  ".join(f"{k}={c}x{ms * 1e3:.0f}us" for k, (c, ms) in top), flush=True)
