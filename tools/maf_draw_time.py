"""k_maf_sample timing at 1M x 32 for the bench's trained MAF proposal and for a randomised (strongly coupled) one;
ASMC_MAF_SAMPLE_ALL_PASSES=1 gives the d-pass loop of round 4 (profiles/r05_maf_draw.txt)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from aspire_amd.engine import HipEngine  # noqa: E402
from aspire_amd.flows import MAFFlow  # noqa: E402

eng = HipEngine(0, n_max=1 << 20, d_max=32)
d, n = 32, 1_000_000
trained = MAFFlow(d, n_transforms=3, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
trained.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)
from conftest import random_maf_flow  # noqa: E402

strong = random_maf_flow(d, 3, 64, seed=4)
for name, flow in (("trained (bench.py's MAF proposal)", trained), ("randomised, strong couplings", strong)):
    dev = flow.device_coupling(eng)
    for mode in ("fixed-point exit", "all d passes"):
        if mode == "all d passes":
            os.environ["ASMC_MAF_SAMPLE_ALL_PASSES"] = "1"
        else:
            os.environ.pop("ASMC_MAF_SAMPLE_ALL_PASSES", None)
        eng.coupling_sample(n, torch.float64, dev, 1, 0, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(5):
            x, lq = eng.coupling_sample(n, torch.float64, dev, 1, 0, 2 + k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"{name:36s} {mode:18s} {ms:7.3f} ms per 1M x 32 draw   nonfinite log q: {int((~torch.isfinite(lq)).sum())}")
