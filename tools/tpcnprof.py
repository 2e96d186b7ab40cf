"""Where a tpCN temperature spends its time (host fit vs kernels) at 1M x 32."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
from aspire_amd.flows import GaussianFlow
from aspire_amd.samplers.smc import HipSMC
from aspire_amd.targets import DiagGaussianMixture

n, d = int(os.environ.get("N", 1_000_000)), 32
eng = HipEngine(0, n_max=n, d_max=32)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
def run(seed, **kw):
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, seed=seed, engine=eng), xp=np,
                engine=eng, rng=np.random.default_rng(2))
    t0 = time.perf_counter()
    post = sp.sample(n, sampler_kwargs=dict(n_steps=32, noise="f32", **kw), store_sample_history=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, sp, post
for i in range(3):
    t, sp, post = run(i)
    print("tpcn run", i, round(t, 4), "s temps", len(sp.history.beta), "logZ", float(post.log_evidence), "nu", [round(v, 1) for v in sp.history.mcmc_nu])
t, sp, post = run(4, step_fn="pcn")
print("pcn run", round(t, 4))
pr = cProfile.Profile(); pr.enable(); run(7); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
