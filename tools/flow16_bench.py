"""Micro-driver of the flow-proposal step above 32 dimensions (k_pcn_flow16 / k_tpcn_flow16) at 1M particles: D=64|128 (or any
32 < d <= 128), KIND=coupling|maf, NU=0|5, STEPS; per-kernel HIP-event table; used under rocprofv3 --pmc by tools/prof_r05.sh."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_coupling_flow, random_maf_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402

n, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 64))
kind, nu, steps = os.environ.get("KIND", "coupling"), float(os.environ.get("NU", 0.0)), int(os.environ.get("STEPS", 8))
eng = HipEngine(0, n_max=n, d_max=128)
w = int(os.environ.get("W", 64))
flow = random_coupling_flow(d, 4, w) if kind == "coupling" else random_maf_flow(d, 3, w)
dev = flow.device_coupling(eng)
g = torch.Generator(eng.device).manual_seed(d)
x = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
t = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
ll, lp, lq = eng.mixture_logpdf(x, t), eng.mixture_logpdf(x, t), eng.coupling_logprob(x, dev)
args = (x, ll, lp, lq, 0.35, mu, eye, eye, t, t, dev, 77, 1000, 0.2, steps, 5, 0.234, False, "f64", nu)
eng.pcn_mutate_flow(*args)
torch.cuda.synchronize()
eng.profile(True)
acc, _, _ = eng.pcn_mutate_flow(*args)
for k, (c, ms) in sorted(eng.profile_report().items(), key=lambda kv: -kv[1][0] * kv[1][1])[:6]:
    print(f"   {k:28s} {c:4d} x {ms * 1e3:9.2f} us")
print(f"{kind} d={d} W={w} nu={nu}: accept {np.mean(acc) / n:.3f}")
