"""Micro-driver of the stand-alone flow density kernels (k_coupling_logprob / k_maf_logprob / k_flow16_logprob) at 1M x D: D, KIND=coupling|maf,
W; per-kernel HIP-event table.  Same-box A/B through ASMC_LIB_PATH."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_coupling_flow, random_maf_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402

n, d, w = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 32)), int(os.environ.get("W", 64))
kind = os.environ.get("KIND", "coupling")
eng = HipEngine(0, n_max=n, d_max=max(d, 32))
flow = random_coupling_flow(d, 4, w) if kind == "coupling" else random_maf_flow(d, 3, w)
dev = flow.device_coupling(eng)
x = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=torch.Generator(eng.device).manual_seed(d))
lq0 = eng.coupling_logprob(x, dev)
torch.cuda.synchronize()
eng.profile(True)
for _ in range(int(os.environ.get("REPS", 20))):
    lq = eng.coupling_logprob(x, dev)
torch.cuda.synchronize()
for k, (c, ms) in sorted(eng.profile_report().items(), key=lambda kv: -kv[1][0] * kv[1][1])[:2]:
    print(f"   {k:24s} {c:4d} x {ms * 1e3:9.2f} us")
print(f"{kind} d={d} W={w}: sum log q = {float(lq.sum()):.10e} (first call {float(lq0.sum()):.10e}), non-finite {int((~torch.isfinite(lq)).sum())}")
