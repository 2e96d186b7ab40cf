"""Per-kernel times of a mutation with the reference's bounded-prior preconditioning (probit of the prior box + affine):
the split path in z-space with asmc_transform_inverse between propose and accept."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd import Aspire, Samples, DiagGaussianMixture
from aspire_amd.engine import HipEngine

n, d = int(os.environ.get("N", 1_000_000)), 32
eng = HipEngine(0, n_max=n, d_max=32)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
bounds = {f"x{i}": [-10.0, 10.0] for i in range(d)}
asp = Aspire(log_likelihood=lik, log_prior=lik, dims=d, parameters=list(bounds), prior_bounds=bounds, flow_backend="gaussian")
asp.fit(Samples(1.5 * np.random.default_rng(0).normal(size=(5000, d)), parameters=list(bounds)))
kw = dict(sampler="smc", n_samples=n, sampler_kwargs=dict(n_steps=8), store_sample_history=False, engine=eng,
          preconditioning="default", preconditioning_kwargs=dict(bounded_to_unbounded=True))
asp.sample_posterior(**{**kw, "n_samples": 65536})
eng.profile(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
post = asp.sample_posterior(**kw)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
rep = eng.profile_report(); eng.profile(False)
print(f"wall {dt:.3f} s  logZ {float(post.log_evidence):.4f} +- {float(post.log_evidence_error):.4f} (analytic {0.5*d*np.log(np.pi) - 0:.4f} up to the box)")
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:30]:
    print(f"  {k:28s} n={v[0]:4d} avg_us={v[1]*1e3:8.1f} total_ms={v[0]*v[1]:8.2f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); asp.sample_posterior(**kw); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
