import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def synth(n, d, seed, sigma_q=1.5):
    """Synthetic Gaussian batch: x = sigma_q N(0,I); ll = lp = -|x|^2/2; lq = log N(0, sigma_q^2 I).
    Same recipe as oracle/make_golden.py (kept separate so the GPU box needs no reference tooling)."""
    g = np.random.default_rng(seed)
    x = sigma_q * g.normal(size=(n, d))
    ll = -0.5 * np.sum(x**2, axis=1)
    lp = ll.copy()
    lq = -0.5 * np.sum((x / sigma_q) ** 2, axis=1) - d * np.log(sigma_q) - 0.5 * d * np.log(2 * np.pi)
    return x, ll, lp, lq


@pytest.fixture(scope="session")
def golden():
    return {name[:-4]: np.load(os.path.join(GOLDEN, name)) for name in os.listdir(GOLDEN) if name.endswith(".npz")}


@pytest.fixture(scope="session")
def oracle():
    import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def hip_engine():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from aspire_amd.engine import HipEngine

    return HipEngine(0, n_max=1 << 21, d_max=128)


def random_coupling_flow(d, n_layers=4, hidden=64, seed=3, dtype=None, device="cpu"):
    """CouplingFlow with every dense layer randomised (the constructor zeroes the output layers) and a
    non-trivial standardisation, for flow parity tests."""
    import torch

    from aspire_amd.flows import CouplingFlow

    flow = CouplingFlow(d, n_layers=n_layers, hidden_features=(hidden, hidden), seed=seed, device=device,
                        dtype=dtype or torch.float32)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for layer in flow.layers:
            for m in layer.net:
                if isinstance(m, torch.nn.Linear):
                    m.weight.copy_((0.5 * torch.randn(m.weight.shape, generator=g) / m.weight.shape[1] ** 0.5).to(m.weight))
                    m.bias.copy_((0.1 * torch.randn(m.bias.shape, generator=g)).to(m.bias))
        flow.loc = (0.3 * torch.randn(d, generator=g)).to(flow.loc)
        flow.scale = (0.5 + torch.rand(d, generator=g)).to(flow.scale)
    flow._version += 1
    return flow


def random_maf_flow(d, n_transforms=3, hidden=64, seed=3):
    """MAFFlow with every (masked) dense layer randomised and a non-trivial standardisation."""
    import torch

    from aspire_amd.flows import MAFFlow, _MaskedLinear

    flow = MAFFlow(d, n_transforms=n_transforms, hidden_features=(hidden, hidden), seed=seed, device="cpu", dtype=torch.float32)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for layer in flow.layers:
            for m in layer.net:
                if isinstance(m, _MaskedLinear):
                    fan = max(1.0, float(m.mask.sum(1).mean()))
                    m.weight.copy_((0.7 * torch.randn(m.weight.shape, generator=g) / fan**0.5).to(m.weight))
                    m.bias.copy_((0.1 * torch.randn(m.bias.shape, generator=g)).to(m.bias))
        flow.loc = (0.3 * torch.randn(d, generator=g)).to(flow.loc)
        flow.scale = (0.5 + torch.rand(d, generator=g)).to(flow.scale)
    flow._version += 1
    return flow


def zuko_like_state_dict(d, hidden=(64, 64), n_transforms=3, seed=0, scale=0.7):
    """A state dict with the key layout and semantics zuko DOCUMENTS for `zuko.flows.MAF(d, 0, transforms=T,
    hidden_features=hidden)` (zuko itself is absent: see MAFFlow.from_zuko_state_dict): per transform a MaskedMLP
    `hyper.{0,2,4}.{weight,bias,mask}` whose last layer emits (shift_i, scale_i) in rows 2 i, 2 i + 1; orders alternate between
    ascending and descending.  Masks are valid MADE masks built from degrees; weights random."""
    g = np.random.default_rng(seed)
    sd = {}
    for i in range(n_transforms):
        order = np.arange(d) if i % 2 == 0 else np.arange(d)[::-1].copy()
        deg = [order]
        for h in hidden:
            deg.append(g.integers(0, max(d - 1, 1), size=h))
        mats = []
        for k, h in enumerate(hidden):
            mats.append((deg[k + 1][:, None] >= deg[k][None, :]).astype(np.float32))
        out_order = np.repeat(order, 2)
        mats.append((out_order[:, None] > deg[-1][None, :]).astype(np.float32))
        for k, m in enumerate(mats):
            fan = max(1.0, m.sum(1).mean())
            sd[f"transform.transforms.{i}.hyper.{2 * k}.weight"] = (scale * g.normal(size=m.shape) / np.sqrt(fan)).astype(np.float32)
            sd[f"transform.transforms.{i}.hyper.{2 * k}.bias"] = (0.1 * g.normal(size=m.shape[0])).astype(np.float32)
            sd[f"transform.transforms.{i}.hyper.{2 * k}.mask"] = m
        sd[f"transform.transforms.{i}.order"] = order.astype(np.int64)
    sd["base._0"] = np.zeros(d, dtype=np.float32)
    sd["base._1"] = np.ones(d, dtype=np.float32)
    return sd


def zuko_documented_log_prob(sd, x):
    """log p(x) of the flow `zuko_like_state_dict` describes, from zuko's documented arithmetic, in numpy fp64: MaskedLinear =
    F.linear(x, mask * weight, bias); ReLU between; phi.unflatten(-1, (-1, 2)) -> (shift, scale);
    MonotonicAffineTransform(shift, scale, slope=1e-3): y = x exp(scale / (1 + |scale / log(slope)|)) + shift; DiagNormal(0, 1) base."""
    x = np.asarray(x, dtype=np.float64)
    n, d = x.shape
    n_tr = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transform.transforms.") and ".hyper." in k)
    z, ladj = x.copy(), np.zeros(n)
    for i in range(n_tr):
        h = z
        for k in (0, 2, 4):
            w = sd[f"transform.transforms.{i}.hyper.{k}.weight"].astype(np.float64) * sd[f"transform.transforms.{i}.hyper.{k}.mask"]
            h = h @ w.T + sd[f"transform.transforms.{i}.hyper.{k}.bias"].astype(np.float64)
            if k < 4:
                h = np.maximum(h, 0.0)
        phi = h.reshape(n, d, 2)
        shift, raw = phi[..., 0], phi[..., 1]
        ls = raw / (1.0 + np.abs(raw / np.log(1e-3)))
        z = z * np.exp(ls) + shift
        ladj += ls.sum(1)
    return -0.5 * (z**2).sum(1) - 0.5 * d * np.log(2 * np.pi) + ladj


def flow_log_prob_f64(flow, x):
    """log q(x) of a CouplingFlow evaluated in fp64 by the torch modules on the CPU: the SAME fp32 parameters, widened
    (the judge of the flow kernels' arithmetic: the north star's bar is 1e-6 relative on log-weights)."""
    import torch

    from aspire_amd.flows import CouplingFlow

    hidden = [m.out_features for m in flow.layers[0].net if isinstance(m, torch.nn.Linear)][:-1]
    f64 = CouplingFlow(flow.dims, n_layers=len(flow.layers), hidden_features=tuple(hidden), device="cpu", dtype=torch.float64)
    f64.layers.load_state_dict({k: v.detach().cpu().double() for k, v in flow.layers.state_dict().items()})
    f64.loc, f64.scale = flow.loc.detach().cpu().double(), flow.scale.detach().cpu().double()
    with torch.no_grad():
        return f64.log_prob(torch.as_tensor(np.asarray(x, dtype=np.float64))).numpy()
