"""asmc_transform_forward / inverse timings at 1M x 32 (bounded -> unbounded + affine)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine
from aspire_amd.transforms import CompositeTransform
n, d = 1_000_000, 32
eng = HipEngine(0, n_max=n, d_max=32)
names = [f"x{i}" for i in range(d)]
for kind in ("probit", "logit", "affine-only"):
    T = CompositeTransform(names, prior_bounds={k: [-10.0, 10.0] for k in names}, bounded_transform=kind if kind != "affine-only" else "probit",
                           bounded_to_unbounded=kind != "affine-only", engine=eng)
    x = torch.rand((n, d), device="cuda", dtype=torch.float64) * 16 - 8
    z = T.fit(x)
    for name, fn, arg in (("forward", T.forward, x), ("inverse", T.inverse, z)):
        for _ in range(3): fn(arg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn(arg)
        torch.cuda.synchronize(); print(kind, name, round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms")
eng.profile(True)
for _ in range(5):
    T.forward(x); T.inverse(z)
print({k: (v[0], round(v[1] * 1e3, 1)) for k, v in eng.profile_report().items()})
