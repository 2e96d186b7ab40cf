"""Where the host spends the idle part of a temperature boundary of the headline run: wall-clock per call of the sampler's
own steps between two mutations (monkeypatched timers; the GPU waits inside them are included and named)."""
import os, sys, time
from collections import defaultdict
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aspire_amd.engine import HipEngine
from aspire_amd.flows import CouplingFlow
from aspire_amd.samplers.smc import HipSMC
from aspire_amd.samples import SMCSamples
from aspire_amd.targets import DiagGaussianMixture
from aspire_amd import smc_math

acc = defaultdict(lambda: [0, 0.0])
def timed(obj, name, label=None):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[label or name]; e[0] += 1; e[1] += time.perf_counter() - t0
    setattr(obj, name, w)

n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
flow.fit(1.35 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)
for name in ("determine_beta", "_stats", "mutate", "_fit_reference", "_upload_reference", "_wrap", "_mutate_steps", "_device_flow"):
    timed(HipSMC, name)
for name in ("resample", "finish_speculation", "speculate_importance_step", "weight_stats"):
    timed(SMCSamples, name)
timed(np.linalg, "cholesky", "np.cholesky"); timed(np.linalg, "inv", "np.inv")
for name in ("pcn_mutate_flow_enqueue", "pcn_mutate_flow_result", "pcn_mutate_flow", "importance_result", "importance_step", "gather", "mean_gram_enqueue", "mean_gram_fetch", "asarray"):
    timed(HipEngine, name, "eng." + name)
def run(seed):
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(seed), dtype="float64")
    return sp, sp.sample(n, sampler_kwargs=dict(n_steps=32, noise="f64", step_fn="pcn"), store_sample_history=False)
run(1); run(2)
acc.clear()
t0 = time.perf_counter(); reps = 5
temps = 0
for k in range(reps):
    sp, _ = run(10 + k); temps += len(sp.history.beta)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"{reps} runs, {temps} temperatures, {1e3 * wall / reps:.2f} ms per run")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} {c:5d} calls  {1e6 * t / temps:9.1f} us per temperature")
