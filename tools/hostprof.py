"""cProfile of one IS-only temperature iteration (host-side overhead hunting)."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd import smc_math
from aspire_amd.comm import Comm
from aspire_amd.engine import HipEngine
from aspire_amd.flows import GaussianFlow
from aspire_amd.samples import gather_global
from aspire_amd.targets import DiagGaussianMixture

n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
comm = Comm()
flow = GaussianFlow(d, sigma=1.5, seed=0, engine=eng)
lik = DiagGaussianMixture.isotropic(d, normalized=False)
x, lq = flow.sample_and_log_prob(n)
ll = eng.mixture_logpdf(x, lik.device_mixture(eng)); lp = ll.clone()
rng = np.random.default_rng(1)

def is_step():
    def eff_fn(betas):
        return [smc_math.ess(s) / n for s in smc_math.global_stats(eng, comm, ll, lp, lq, 0.0, betas, n)]
    def search_fn(b0, target, tol):
        b, _, conv, passes, n_nan = eng.find_beta(ll, lp, lq, b0, target, tol)[:5]
        return b, passes
    beta, _, _ = smc_math.determine_beta(eff_fn, 0.0, adaptive=True, beta_step=float("nan"), min_beta_step=0.0, max_beta_step=1.0,
                                         beta_tolerance=1e-6, adaptive_min_beta_step=False, target=0.5, rate=1.0,
                                         search_fn=None if os.environ.get("HOSTBIS") else search_fn)
    st_b, st_1 = smc_math.global_stats(eng, comm, ll, lp, lq, 0.0, [beta, 1.0], n)
    smc_math.evidence_variance(eng, comm, ll, lp, lq, 0.0, beta, st_b)
    idx, _ = smc_math.resample_indices(eng, comm, ll, lp, lq, 0.0, beta, n, rng, mode=os.environ.get("MODE", "exact"))
    return gather_global(eng, comm, idx, x, ll, lp, lq)

for _ in range(3): is_step()
torch.cuda.synchronize()
for blk in range(8):
    t0 = time.perf_counter()
    for _ in range(10): is_step()
    torch.cuda.synchronize()
    print("block", blk, "ms/step", (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): is_step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

# per-step wall times + per-call outliers
import collections
slow = collections.Counter()
orig = {}
for name in ["weights_stats", "weights_sums", "weights_m2", "normalized_weights", "cdf", "cdf_normalize", "uniforms_pcg64", "search", "gather"]:
    fn = getattr(eng, name); orig[name] = fn
    def w(*a, _fn=fn, _n=name, **k):
        t = time.perf_counter(); r = _fn(*a, **k); dt = time.perf_counter() - t
        if dt > 2e-3: slow[_n] += 1; print("slow call", _n, round(dt * 1e3, 2), "ms")
        return r
    setattr(eng, name, w)
ts = []
for i in range(60):
    t0 = time.perf_counter(); is_step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("steps >3ms:", [(i, round(t, 1)) for i, t in enumerate(ts) if t > 3], "median", sorted(ts)[30])
