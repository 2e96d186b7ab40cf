"""A/B of the split conversion (v_fma_mix form against the cvt + subtract form, -DFLOW_SPLIT_CVT): the stand-alone flow kernel's
log q on the same inputs must be bit-identical - in-range rows, far-out rows and rows small enough that the lo halves are
fp16 subnormals.  usage: mix_check.py dump <out.npy>   (ASMC_LIB_PATH selects the library)   |   mix_check.py cmp a.npy b.npy"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = a.view(np.int64) == b.view(np.int64)
    print(f"{a.size} densities, bit-identical: {int(same.sum())} ({'ALL' if same.all() else 'MISMATCH'}); finite {int(np.isfinite(a).sum())}")
    if not same.all():
        i = np.flatnonzero(~same)[:5]
        print(i, a[i], b[i])
        sys.exit(1)
    sys.exit(0)

import torch  # noqa: E402
from conftest import random_coupling_flow  # noqa: E402

from aspire_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0, n_max=1 << 18, d_max=32)
outs = []
for hidden, layers in ((64, 4), (32, 4), (128, 1)):
    flow = random_coupling_flow(32, layers, hidden)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(hidden)
    for scale in (1.0, 4.0, 1e-2, 1e-4, 1e-6):
        x = scale * torch.randn((1 << 16, 32), device=eng.device, dtype=torch.float64, generator=g)
        outs.append(eng.coupling_logprob(x, dev).cpu().numpy())
np.save(sys.argv[2], np.concatenate(outs))
print("dumped", sum(o.size for o in outs))
