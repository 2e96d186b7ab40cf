import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aspire_amd.engine import HipEngine
from aspire_amd.flows import GaussianFlow
from aspire_amd.samplers.smc import HipSMC
from aspire_amd.targets import DiagGaussianMixture
n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
def run(seed, prof=False, **kw):
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, seed=seed, engine=eng), xp=np, engine=eng, rng=np.random.default_rng(2))
    if prof: eng.profile(True)
    t0 = time.perf_counter()
    sp.sample(n, sampler_kwargs=dict(n_steps=32, noise="f32", **kw), store_sample_history=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rep = eng.profile_report() if prof else None
    if prof: eng.profile(False)
    return dt, rep
run(0); run(1)
dt, rep = run(2, prof=True)
print("tpcn", dt)
tot = 0
for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    tot += c * ms
    print(f"  {k:28s} {c:5d} x {ms * 1e3:8.2f} us = {c * ms:7.3f} ms")
print("sum", tot)
