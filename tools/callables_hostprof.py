"""Host-side profile (cProfile) of a mutation with Python densities at 1M x 32: where the interpreter time of a step goes."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
from aspire_amd.flows import GaussianFlow
from aspire_amd.history import SMCHistory
from aspire_amd.samplers.smc import HipSMC
from aspire_amd.samples import SMCSamples

n, d, steps = int(os.environ.get("N", 1_000_000)), 32, int(os.environ.get("STEPS", 32))
eng = HipEngine(0, n_max=n, d_max=32)
tlik = lambda smp: -0.5 * (smp.x * smp.x).sum(1)  # noqa: E731
flow = GaussianFlow(d, sigma=1.5, seed=1, engine=eng)
sp = HipSMC(log_likelihood=tlik, log_prior=tlik, dims=d, prior_flow=flow, xp=torch, engine=eng, rng=np.random.default_rng(5))
sp.sampler_kwargs = dict(n_steps=steps, step_fn=os.environ.get("STEP_FN", "pcn"))
sp.history = SMCHistory()
x, lq = flow.sample_and_log_prob(n)
ll = -0.5 * (x * x).sum(1)
pop = SMCSamples(x=x, log_likelihood=ll, log_prior=ll.clone(), log_q=lq, beta=0.3, xp=torch, engine=eng)
sp._pcn_state = {"rho": None, "step": 0, "nu": None}
sp.adaptive, sp.device_bisection = True, True
sp.mutate(pop, 0.3)
torch.cuda.synchronize()
t0 = time.perf_counter()
sp.mutate(pop, 0.3)
torch.cuda.synchronize()
print("wall per step %.3f ms [%s]" % ((time.perf_counter() - t0) / steps * 1e3, sp.last_mutation_path))
pr = cProfile.Profile()
pr.enable()
sp.mutate(pop, 0.3)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
