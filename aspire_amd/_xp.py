"""Minimal array-namespace helpers (numpy / torch) for the host-side containers.

The reference routes every array through array_api_compat (src/aspire/utils.py:258-315); here only
numpy and torch exist: numpy for host-facing results, torch for device-resident particle state.
"""
from __future__ import annotations

import numpy as np
import torch


def is_torch(x) -> bool:
    return isinstance(x, torch.Tensor)


def is_torch_namespace(xp) -> bool:
    return xp is torch or getattr(xp, "__name__", "").endswith("torch")


def namespace_of(x):
    return torch if is_torch(x) else np


def to_numpy(x) -> np.ndarray:
    """utils.py:258-275: detach + move to the CPU."""
    if is_torch(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)


_TORCH_DT = {"float32": torch.float32, "float64": torch.float64}
_NP_DT = {torch.float32: np.float32, torch.float64: np.float64}


def resolve_dtype(dtype, xp):
    """str / numpy / torch dtype -> dtype object of namespace xp (utils.py resolve_dtype)."""
    if dtype is None:
        return None
    if is_torch_namespace(xp):
        if isinstance(dtype, torch.dtype):
            return dtype
        return _TORCH_DT[np.dtype(dtype).name]
    if isinstance(dtype, torch.dtype):
        return np.dtype(_NP_DT[dtype])
    return np.dtype(dtype)


def resolve_xp(name, default=np):
    """The array namespace a saved file NAMES (reference utils.py resolve_xp: a backend name -> its module).  Only array namespaces are
    accepted - `numpy`, `torch`, and the `array_api_compat` wrappers of the two wherever that package lives (`array_api_compat.torch`,
    `sklearn.externals.array_api_compat.numpy`, ...): a file is data and does not get to import arbitrary modules.  A wrapper package that
    is not installed resolves to the plain module; anything else to `default`."""
    import importlib

    if not isinstance(name, str) or not name:
        return default
    last = name.rsplit(".", 1)[-1]
    if last not in ("numpy", "torch") or not (name == last or name.endswith("array_api_compat." + last)):
        return default
    try:
        return importlib.import_module(name)
    except ImportError:
        return torch if last == "torch" else np


def default_dtype(xp):
    return torch.get_default_dtype() if is_torch_namespace(xp) else np.dtype(np.float64)


def asarray(x, xp, dtype=None, device=None):
    """utils.py:278-315 restricted to numpy <-> torch."""
    if is_torch_namespace(xp):
        dt = resolve_dtype(dtype, torch)
        if is_torch(x):
            return x.to(device=device if device is not None else x.device, dtype=dt if dt is not None else x.dtype)
        return torch.as_tensor(np.asarray(x), dtype=dt, device=device)
    dt = resolve_dtype(dtype, np)
    a = to_numpy(x)
    return a if dt is None else a.astype(dt, copy=False)
