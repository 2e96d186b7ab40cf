"""Student-t reference distribution of the t-preconditioned Crank-Nicolson step (`step_fn="tpcn"`, the reference's
default: smc/minipcn.py:46-49; the fit itself lives in third-party minipcn, absent => this repository's specification).

`fit_student_t` is the standard EM for a multivariate t with unknown location, scale matrix and degrees of freedom
(Liu & Rubin 1995): E-step weights z_i = (nu + d)/(nu + delta_i), delta_i the Mahalanobis distance; M-step weighted
mean and scale matrix; nu solves  -psi(nu/2) + log(nu/2) + 1 + mean(log z_i - z_i) + psi((nu+d)/2) - log((nu+d)/2) = 0.
It runs over a SUBSAMPLE of the particles: the reference distribution only shapes the proposal - the Metropolis
correction keeps the tempered target invariant for any (mu, Sigma, nu) - so a few thousand particles are plenty.
`fit_student_t_device` is what the sampler uses: everything that touches a particle (Mahalanobis distances, weights,
weighted sums, the weighted scatter matrix) runs in HIP kernels on the device-resident subsample
(asmc_student_estep / asmc_student_scale / asmc_centered_gram); the host keeps the d-vector, d x d and scalar algebra
(weighted mean, Cholesky factor, the root in nu), as it does for the Gaussian reference.  `fit_student_t` is the same
EM in numpy (test reference; also the restatement the device version is checked against).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import brentq
from scipy.special import digamma

NU_MIN, NU_MAX = 1.0, 1.0e6
NU_GAUSSIAN = 1.0e5  # above this the Student-t reference is numerically the Gaussian one


def _chol(cov: np.ndarray) -> np.ndarray:
    d = cov.shape[0]
    scale = float(np.mean(np.diag(cov)))
    if not np.isfinite(scale) or scale <= 0:
        scale = 1.0
    jitter = 0.0
    for _ in range(12):
        try:
            return np.linalg.cholesky(cov + jitter * scale * np.eye(d))
        except np.linalg.LinAlgError:
            jitter = 1e-12 if jitter == 0.0 else jitter * 100
    raise RuntimeError("could not factor the scale matrix of the Student-t fit")


def fit_student_t(data: np.ndarray, max_iter: int = 50, rtol: float = 1e-3, nu0: float = 20.0):
    """(mu [d], Sigma [d, d], nu) of the multivariate Student-t fitted to the rows of `data` by EM."""
    data = np.asarray(data, dtype=np.float64)
    n, d = data.shape
    mu = data.mean(axis=0)
    cov = np.atleast_2d(np.cov(data.T)) if n > 1 else np.eye(d)
    cov = 0.5 * (cov + cov.T)
    nu = float(nu0)
    for _ in range(max_iter):
        Linv = np.linalg.inv(_chol(cov))  # d x d
        y = (data - mu) @ Linv.T
        delta = np.einsum("nj,nj->n", y, y)
        z = (nu + d) / (nu + delta)
        mu = (z[:, None] * data).sum(axis=0) / z.sum()
        diff = data - mu
        cov = (diff * z[:, None]).T @ diff / n
        cov = 0.5 * (cov + cov.T)
        c = 1.0 + float(np.mean(np.log(z) - z)) + float(digamma(0.5 * (nu + d))) - np.log(0.5 * (nu + d))

        def f(v):
            return -digamma(0.5 * v) + np.log(0.5 * v) + c

        # f decreases from +inf (v -> 0) to c (v -> inf): a root exists iff c < 0; otherwise the data are Gaussian
        if not (c < 0.0) or f(NU_MAX) >= 0.0:
            nu_new = NU_MAX
        elif f(NU_MIN) <= 0.0:
            nu_new = NU_MIN
        else:
            nu_new = float(brentq(f, NU_MIN, NU_MAX, xtol=1e-8, rtol=1e-10))
        done = abs(nu_new - nu) <= rtol * nu
        nu = nu_new
        if done:
            break
    return mu, cov, float(min(max(nu, NU_MIN), NU_MAX))


def fit_student_t_device(engine, xs, max_iter: int = 50, rtol: float = 1e-3, nu0: float = 20.0):
    """`fit_student_t` with the per-particle arithmetic on the device.  xs: [m, d] fp64 device tensor."""
    m, d = int(xs.shape[0]), int(xs.shape[1])
    mu = np.asarray(engine.colsum(xs), dtype=np.float64) / m
    cov = np.asarray(engine.centered_gram(xs, mu), dtype=np.float64) / max(m - 1, 1)
    cov = 0.5 * (cov + cov.T)
    nu = float(nu0)
    zero = np.zeros(d)
    for _ in range(max_iter):
        Linv = np.linalg.inv(_chol(cov))
        z, sum_z, sum_lz, sum_zx = engine.student_estep(xs, mu, np.tril(Linv), nu)
        mu = sum_zx / sum_z
        cov = np.asarray(engine.centered_gram(engine.student_scale(xs, z, mu), zero), dtype=np.float64) / m
        cov = 0.5 * (cov + cov.T)
        c = 1.0 + sum_lz / m + float(digamma(0.5 * (nu + d))) - np.log(0.5 * (nu + d))

        def f(v):
            return -digamma(0.5 * v) + np.log(0.5 * v) + c

        if not (c < 0.0) or f(NU_MAX) >= 0.0:
            nu_new = NU_MAX
        elif f(NU_MIN) <= 0.0:
            nu_new = NU_MIN
        else:
            nu_new = float(brentq(f, NU_MIN, NU_MAX, xtol=1e-8, rtol=1e-10))
        done = abs(nu_new - nu) <= rtol * nu
        nu = nu_new
        if done:
            break
    return mu, cov, float(min(max(nu, NU_MIN), NU_MAX))
