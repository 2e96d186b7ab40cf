import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
x = torch.randn((n, d), device="cuda", dtype=torch.float64)
tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
ll = eng.mixture_logpdf(x, tgt); lp = ll.clone(); lq = ll.clone()
mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
def T(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("colsum ms", T(lambda: eng.colsum(x)))
mean = eng.colsum(x) / n
print("gram ms", T(lambda: eng.centered_gram(x, mean)))
print("pcn32 ms", T(lambda: eng.pcn_mutate(x, ll, lp, lq, 0.5, mu, eye, eye, tgt, tgt, tgt, 1, 0, 0.3, 32, 0, 0.234, True, "f32")))
print("pcn1 ms", T(lambda: eng.pcn_mutate(x, ll, lp, lq, 0.5, mu, eye, eye, tgt, tgt, tgt, 1, 0, 0.3, 1, 0, 0.234, True, "f32")))
print("stats15 ms", T(lambda: eng.weights_stats(ll, lp, lq, 0.0, np.linspace(0.1, 0.9, 15))))

# sampler-like sequence with per-phase wall times
from aspire_amd import smc_math
from aspire_amd.comm import Comm
from aspire_amd.samples import gather_global
rng = np.random.default_rng(0)
def phase(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print(f"   {name:10s} {(time.perf_counter()-t0)*1e3:8.3f} ms"); return r
xs, a, b, c = x, ll, lp, lq
for it in range(4):
    print("iteration", it)
    idx = phase("indices", lambda: smc_math.resample_indices(eng, Comm(), a, b, c, 0.0, 0.05, n, rng, mode="exact")[0])
    xs, a, b, c = phase("gather", lambda: gather_global(eng, Comm(), idx, xs, a, b, c))
    s = phase("colsum", lambda: eng.colsum(xs))
    g = phase("gram", lambda: eng.centered_gram(xs, s / n))
    phase("pcn32", lambda: eng.pcn_mutate(xs, a, b, c, 0.5, mu, eye, eye, tgt, tgt, tgt, 1, 0, 0.3, 32, 0, 0.234, True, "f32"))
