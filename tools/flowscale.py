#!/usr/bin/env python
"""Flow-kernel time vs particle count (separates fixed launch/fill cost from per-tile cost)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_coupling_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402
from tools.kbench import timeit  # noqa: E402

d = 32
eng = HipEngine(0, n_max=1 << 23, d_max=32)
flow = random_coupling_flow(d, 4, 64)
dev = flow.device_coupling(eng)
for n in (65536, 131072, 262144, 524288, 1000000, 2097152, 4194304, 8388608):
    for xdt in (torch.float32,):
        x = torch.randn((n, d), device="cuda", dtype=xdt)
        ms = timeit(lambda: eng.coupling_logprob(x, dev), reps=5, warm=2)
        flops = n * 4 * 2 * ((d // 2) * 64 + 64 * 64 + 64 * d)
        print(f"n={n:8d} {ms*1e3:8.1f} us  {flops/ms/1e9:6.1f} TFLOP/s  {ms*1e6/n*1e3:.1f} ps/particle")
