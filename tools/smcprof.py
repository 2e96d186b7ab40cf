"""cProfile of a full HipSMC.sample() run at 1M x 32 (host-side overhead hunting)."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
from aspire_amd.flows import GaussianFlow
from aspire_amd.samplers.smc import HipSMC
from aspire_amd.targets import DiagGaussianMixture

n, d = int(os.environ.get("N", 1_000_000)), 32
eng = HipEngine(0, n_max=n, d_max=32)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
def run(seed):
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, seed=seed, engine=eng), xp=np,
                engine=eng, rng=np.random.default_rng(2))
    t0 = time.perf_counter()
    post = sp.sample(n, sampler_kwargs=dict(n_steps=32, noise="f32"), store_sample_history=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, sp, post
for i in range(3):
    t, sp, post = run(i)
    print("run", i, round(t, 4), "s temps", len(sp.history.beta), "logZ", float(post.log_evidence))
eng.profile(True)
t, sp, post = run(5)
rep = eng.profile_report(); eng.profile(False)
tot = sum(c * ms for c, ms in rep.values())
print("profiled run", round(t, 4), "s; kernel time total ms", round(tot, 2))
for k, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:12]:
    print(f"  {k:28s} n={c:5d} avg_us={ms*1e3:9.2f} total_ms={c*ms:8.2f}")
pr = cProfile.Profile(); pr.enable(); run(7); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# wall time per engine call inside the real sampler (synchronised before/after each call)
import collections
acc = collections.defaultdict(lambda: [0, 0.0])
for name in dir(eng):
    if name.startswith("_") or name in ("profile", "profile_report", "asarray", "empty", "full", "to_numpy", "synchronize", "close", "ensure_capacity", "make_mixture"):
        continue
    fn = getattr(eng, name)
    if not callable(fn):
        continue
    def w(*a, _fn=fn, _n=name, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = _fn(*a, **k); torch.cuda.synchronize()
        acc[_n][0] += 1; acc[_n][1] += time.perf_counter() - t0
        return r
    setattr(eng, name, w)
t, sp, post = run(9)
print("instrumented run", round(t, 4), "s")
for k, (c, tt) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:22s} n={c:4d} total_ms={tt*1e3:9.2f} avg_ms={tt/c*1e3:8.3f}")
print("sum ms", sum(v[1] for v in acc.values()) * 1e3)
