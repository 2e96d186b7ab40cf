#!/usr/bin/env python
"""Per-kernel times of the moments step (colsum + centred Gram) at N x D."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine  # noqa: E402

n, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 128))
eng = HipEngine(0, n_max=n, d_max=max(d, 32))
x = torch.randn((n, d), device="cuda", dtype=torch.float64)
mean = eng.colsum(x) / n
eng.centered_gram(x, mean)
eng.profile(True)
for _ in range(5):
    eng.colsum(x)
    eng.centered_gram(x, mean)
for k, v in eng.profile_report().items():
    print(f"{k:24s} n={v[0]:3d} avg_us={v[1]*1e3:9.1f}")
