"""Built-in device-resident log-densities for `log_likelihood` / `log_prior`.

The reference takes arbitrary Python callables `f(samples) -> array` (src/aspire/samplers/base.py:81-91).
Those still work here (split propose/accept path), but a callable that is an instance of
`DiagGaussianMixture` is *recognised* by the sampler and evaluated inside the fused pCN kernel, so
the particle state never leaves HBM.  Called directly it behaves like any other
user callable and evaluates the same formula with plain array ops of the samples' namespace.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ._xp import is_torch


class DiagGaussianMixture:
    """log sum_c w_c N(x; mean_c, diag var_c)  (or the un-normalised form).

    Parameters
    ----------
    means : [C, d] or [d]
    variances : [C, d], [d] or scalar
    weights : [C] mixture weights (default equal)
    normalized : include the -d/2 log(2 pi) - 1/2 sum log var terms (default True)
    log_scale : constant added to the log-density
    """

    def __init__(self, means, variances=1.0, weights=None, normalized: bool = True, log_scale: float = 0.0):
        mu = np.atleast_2d(np.asarray(means, dtype=np.float64))
        C, d = mu.shape
        var = np.broadcast_to(np.asarray(variances, dtype=np.float64), (C, d)).copy()
        if np.any(var <= 0):
            raise ValueError("variances must be positive")
        w = np.full(C, 1.0 / C) if weights is None else np.asarray(weights, dtype=np.float64)
        if w.shape != (C,) or np.any(w <= 0):
            raise ValueError("weights must be C positive numbers")
        self.mu, self.prec = mu, 1.0 / var
        logw = np.log(w / w.sum()) + log_scale
        if normalized:
            logw = logw - 0.5 * d * math.log(2 * math.pi) - 0.5 * np.sum(np.log(var), axis=1)
        self.logw = logw
        self.dims = d
        self._dev = {}

    @classmethod
    def isotropic(cls, dims: int, mean: float = 0.0, var: float = 1.0, normalized: bool = True):
        return cls(np.full((1, dims), mean), var, normalized=normalized)

    def device_mixture(self, engine):
        key = id(engine)
        if key not in self._dev:
            self._dev[key] = engine.make_mixture(self.logw, self.mu, self.prec)
        return self._dev[key]

    def __call__(self, samples):
        x = samples.x if hasattr(samples, "x") else samples
        if is_torch(x):
            mu = torch.as_tensor(self.mu, dtype=torch.float64, device=x.device)
            pr = torch.as_tensor(self.prec, dtype=torch.float64, device=x.device)
            lw = torch.as_tensor(self.logw, dtype=torch.float64, device=x.device)
            t = x.to(torch.float64)[:, None, :] - mu[None]
            terms = lw[None] - 0.5 * (t * t * pr[None]).sum(-1)
            return terms[:, 0] if terms.shape[1] == 1 else torch.logsumexp(terms, dim=1)
        x = np.asarray(x, dtype=np.float64)
        t = x[:, None, :] - self.mu[None]
        terms = self.logw[None] - 0.5 * (t * t * self.prec[None]).sum(-1)
        if terms.shape[1] == 1:
            return terms[:, 0]
        m = terms.max(axis=1)
        with np.errstate(invalid="ignore"):
            out = m + np.log(np.exp(terms - m[:, None]).sum(axis=1))
        return np.where(np.isneginf(m), -np.inf, out)
