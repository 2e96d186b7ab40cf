"""Mutation with arbitrary (torch) callables as the densities at several d: wall time per step of `HipSMC.mutate`, the code
path it took and the kernels behind it.  N, STEPS env."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import GaussianFlow  # noqa: E402
from aspire_amd.samplers.smc import HipSMC  # noqa: E402
from aspire_amd.samples import SMCSamples  # noqa: E402

n, steps = int(os.environ.get("N", 1_000_000)), int(os.environ.get("STEPS", 8))
eng = HipEngine(0, n_max=n, d_max=128)
for d in [int(v) for v in os.environ.get("DIMS", "16,20,32,48,64,100,128").split(",")]:
    tlik = lambda smp: -0.5 * (smp.x * smp.x).sum(1)  # noqa: E731
    flow = GaussianFlow(d, sigma=1.5, seed=1, engine=eng)
    sp = HipSMC(log_likelihood=tlik, log_prior=tlik, dims=d, prior_flow=flow, xp=torch, engine=eng, rng=np.random.default_rng(5))
    sp.sampler_kwargs = dict(n_steps=steps, step_fn=os.environ.get("STEP_FN", "pcn"))
    from aspire_amd.history import SMCHistory

    sp.history = SMCHistory()
    x, lq = flow.sample_and_log_prob(n)
    ll = -0.5 * (x * x).sum(1)
    pop = SMCSamples(x=x, log_likelihood=ll, log_prior=ll.clone(), log_q=lq, beta=0.3, xp=torch, engine=eng)
    sp._pcn_state = {"rho": None, "step": 0, "nu": None}
    sp.adaptive, sp.device_bisection = True, True
    sp.mutate(pop, 0.3)
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    sp.mutate(pop, 0.3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rep = eng.profile_report()
    eng.profile(False)
    top = sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:5]
    print(f"d={d:3d}: {dt / steps * 1e3:7.3f} ms/step  [{sp.last_mutation_path}]  " + "  ".join(f"{k}={c}x{ms * 1e3:.0f}us" for k, (c, ms) in top))
