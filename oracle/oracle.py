"""ctypes front-end of the CPU oracle (`oracle/asmc_oracle.c`).

TEST INFRASTRUCTURE — the checker, never the product.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
`aspire_amd/` must never import it (tests/test_layout.py enforces that).

Parity: pinned for the weights / ESS / bisection / evidence / resample functions (golden vectors
from the real reference, `tests/golden/ref_*.npz`); UNPINNED for the pCN functions (third-party
`minipcn`, absent — see asmc_oracle.c header).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libasmc_oracle.so")

ORC_OK, ORC_ERR_ARG, ORC_ERR_NAN, ORC_ERR_PSUM, ORC_ERR_BETA_STALL = 0, -1, -2, -3, -4


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "asmc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


class _Mixture(ctypes.Structure):
    _fields_ = [
        ("C", ctypes.c_int),
        ("logw", ctypes.c_void_p),
        ("mu", ctypes.c_void_p),
        ("prec", ctypes.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        d, i64, vp, ci = ctypes.c_double, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        u64, u32 = ctypes.c_uint64, ctypes.c_uint32
        sig = {
            "orc_logsumexp": (d, [vp, i64]),
            "orc_unnormalized_log_weights": (None, [i64, vp, vp, vp, d, d, vp]),
            "orc_log_weights": (ci, [i64, vp, vp, vp, d, d, vp]),
            "orc_effective_sample_size": (d, [vp, i64]),
            "orc_ess_at_beta": (d, [i64, vp, vp, vp, d, d, vp]),
            "orc_log_evidence_ratio": (d, [i64, vp, vp, vp, d, d]),
            "orc_log_evidence_ratio_variance": (d, [i64, vp, vp, vp, d, d]),
            "orc_current_target_efficiency": (d, [d, ci, d, d, d]),
            "orc_determine_beta": (ci, [i64, vp, vp, vp, d, d, d, d, d, ci, ci, ci, d, d, d, vp]),
            "orc_pcg64_advance": (None, [vp, u64, u64]),
            "orc_pcg64_random": (None, [vp, i64, vp]),
            "orc_cdf_from_weights": (None, [i64, vp, vp]),
            "orc_searchsorted_right": (None, [i64, vp, i64, vp, vp]),
            "orc_resample_indices": (ci, [i64, vp, vp, vp, d, d, ci, i64, vp, vp, vp]),
            "orc_normalized_weights": (ci, [i64, vp, vp, vp, d, d, vp]),
            "orc_gather_rows": (None, [i64, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
            "orc_systematic_uniforms": (None, [i64, d, vp]),
            "orc_stratified_uniforms": (None, [i64, vp, vp]),
            "orc_log_p_t": (d, [d, d, d, d]),
            "orc_compact_valid": (i64, [i64, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
            "orc_philox4x32_10": (None, [vp, vp, vp]),
            "orc_pcn_noise": (None, [u64, u64, u32, ci, vp, vp]),
            "orc_diag_mixture_logpdf": (d, [ci, ci, vp, vp, vp, vp]),
            "orc_pcn_step": (
                i64,
                [i64, ci, vp, vp, vp, vp, d, vp, vp, vp, d, vp, vp, vp, u64, u64, u32, ci],
            ),
            "orc_pcn_noise_f32": (None, [u64, u64, u32, ci, vp, vp]),
            "orc_gamma_unit": (d, [d, u64, u64, u32]),
            "orc_gamma_unit_mode": (d, [d, u64, u64, u32, ctypes.c_int]),
            "orc_tpcn_corr": (d, [d, ci, d]),
            "orc_tpcn_step": (
                i64,
                [i64, ci, vp, vp, vp, vp, d, vp, vp, vp, d, d, vp, vp, vp, u64, u64, u32, ci],
            ),
            "orc_pcn_adapt": (d, [d, d, d, ci]),
            "orc_moments": (None, [i64, ci, vp, vp, vp]),
            "orc_is_iteration": (ci, [i64, ci, vp, vp, vp, vp, d, d, d, vp, vp, vp, vp, vp, vp]),
            "orc_coupling_logprob": (ci, [i64, ci, vp, ci, ci, vp, vp, vp, vp, vp]),
            "orc_maf_logprob": (ci, [i64, ci, vp, ci, ci, vp, vp, vp, vp, vp]),
            "orc_maf_logprob_form": (ci, [i64, ci, vp, ci, ci, vp, vp, vp, vp, ci, vp]),
            "orc_transform": (ci, [i64, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, d, ci]),
            "orc_pcn_flow_step": (
                i64,
                [i64, ci, vp, vp, vp, vp, d, vp, vp, vp, d, vp, vp, ci, ci, vp, vp, vp, vp, u64, u64, u32, ci, ci],
            ),
            "orc_pcn_flow_step_kind": (
                i64,
                [i64, ci, vp, vp, vp, vp, d, vp, vp, vp, d, vp, vp, ci, ci, vp, vp, vp, vp, u64, u64, u32, ci, ci, ci],
            ),
            "orc_tpcn_flow_step_kind": (
                i64,
                [i64, ci, vp, vp, vp, vp, d, vp, vp, vp, d, d, vp, vp, ci, ci, vp, vp, vp, vp, u64, u64, u32, ci, ci, ci],
            ),
            "orc_max_threads": (ci, []),
            "orc_set_margin_sink": (None, [vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data


class OracleNaNError(ValueError):
    pass


def _check(st, beta=None):
    if st == ORC_ERR_NAN:
        # same exception type/message as reference samples.py:1246-1247
        raise ValueError(f"Log weights contain NaN values for beta={beta}")
    if st == ORC_ERR_PSUM:
        raise ValueError("probabilities do not sum to 1")
    if st not in (ORC_OK,):
        raise RuntimeError(f"oracle error {st}")


def logsumexp(x):
    x = _f64(x)
    return lib().orc_logsumexp(_p(x), x.size)


def unnormalized_log_weights(ll, lp, lq, beta0, beta):
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    out = np.empty_like(ll)
    lib().orc_unnormalized_log_weights(ll.size, _p(ll), _p(lp), _p(lq), beta0, beta, _p(out))
    return out


def log_weights(ll, lp, lq, beta0, beta):
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    out = np.empty_like(ll)
    _check(lib().orc_log_weights(ll.size, _p(ll), _p(lp), _p(lq), beta0, beta, _p(out)), beta)
    return out


def effective_sample_size(lw):
    lw = _f64(lw)
    return lib().orc_effective_sample_size(_p(lw), lw.size)


def ess_at_beta(ll, lp, lq, beta0, beta):
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    st = ctypes.c_int(0)
    r = lib().orc_ess_at_beta(ll.size, _p(ll), _p(lp), _p(lq), beta0, beta, ctypes.addressof(st))
    _check(st.value, beta)
    return r


def log_evidence_ratio(ll, lp, lq, beta0, beta):
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    return lib().orc_log_evidence_ratio(ll.size, _p(ll), _p(lp), _p(lq), beta0, beta)


def log_evidence_ratio_variance(ll, lp, lq, beta0, beta):
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    return lib().orc_log_evidence_ratio_variance(ll.size, _p(ll), _p(lp), _p(lq), beta0, beta)


@dataclass
class BetaResult:
    beta: float
    min_beta_step: float
    beta_star: float
    n_evals: int
    stalled: bool


def determine_beta(
    ll,
    lp,
    lq,
    beta,
    beta_step=float("nan"),
    min_beta_step=0.0,
    max_beta_step=1.0,
    beta_tolerance=1e-8,
    adaptive=True,
    adaptive_min_beta_step=False,
    target_efficiency=0.5,
    target_efficiency_rate=1.0,
) -> BetaResult:
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    if isinstance(target_efficiency, (tuple, list)):
        at, t0, t1 = 1, float(target_efficiency[0]), float(target_efficiency[1])
    else:
        at, t0, t1 = 0, float(target_efficiency), 0.0
    out = np.zeros(4)
    st = lib().orc_determine_beta(
        ll.size, _p(ll), _p(lp), _p(lq), beta, beta_step, min_beta_step, max_beta_step,
        beta_tolerance, int(adaptive), int(adaptive_min_beta_step), at, t0, t1,
        float(target_efficiency_rate), _p(out),
    )
    if st == ORC_ERR_BETA_STALL:
        return BetaResult(out[0], out[1], out[2], int(out[3]), True)
    _check(st)
    return BetaResult(out[0], out[1], out[2], int(out[3]), False)


def pcg64_state_from_numpy(rng) -> np.ndarray:
    """{state_hi, state_lo, inc_hi, inc_lo} of a numpy Generator backed by PCG64."""
    st = rng.bit_generator.state
    if st["bit_generator"] != "PCG64":
        raise TypeError("need a PCG64 generator")
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m], dtype=np.uint64)


def pcg64_random(state: np.ndarray, n: int) -> np.ndarray:
    out = np.empty(n)
    lib().orc_pcg64_random(_p(state), n, _p(out))
    return out


def pcg64_advance(state: np.ndarray, delta: int) -> None:
    m = (1 << 64) - 1
    lib().orc_pcg64_advance(_p(state), (delta >> 64) & m, delta & m)


SELECT_THREADS = 262144  # include/asmc.h ASMC_SELECT_THREADS


def pcg64_select(state: np.ndarray, n_total: int, lo: float, hi: float) -> np.ndarray:
    """Specification of asmc_pcg64_select (this repository's sharded owner-layout resampling; no reference
    counterpart): of the next n_total doubles of the stream keep lo <= u < hi, map to q = (u - lo) * (1 / (hi - lo))
    clamped below 1, ordered by (thread // 64, iteration, thread % 64) with thread = i % 262144, iteration = i // 262144."""
    st = np.array(state, dtype=np.uint64)
    u = pcg64_random(st, int(n_total))
    i = np.arange(int(n_total), dtype=np.int64)
    j, k = i % SELECT_THREADS, i // SELECT_THREADS
    order = np.lexsort((j % 64, k, j // 64))
    uo = u[order]
    uo = uo[(uo >= lo) & (uo < hi)]
    q = (uo - lo) * (1.0 / (hi - lo))
    return np.where(q >= 1.0, np.nextafter(1.0, 0.0), q)


def cdf_from_weights(w):
    w = _f64(w)
    cdf = np.empty_like(w)
    lib().orc_cdf_from_weights(w.size, _p(w), _p(cdf))
    return cdf


def searchsorted_right(cdf, u):
    cdf, u = _f64(cdf), _f64(u)
    idx = np.empty(u.size, dtype=np.int64)
    lib().orc_searchsorted_right(cdf.size, _p(cdf), u.size, _p(u), _p(idx))
    return idx


def normalized_weights(ll, lp, lq, beta0, beta):
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    w = np.empty_like(ll)
    _check(lib().orc_normalized_weights(ll.size, _p(ll), _p(lp), _p(lq), beta0, beta, _p(w)), beta)
    return w


def resample_indices(ll, lp, lq, beta0, beta, u, uniform_weights=False, return_cdf=False):
    ll, lp, lq, u = _f64(ll), _f64(lp), _f64(lq), _f64(u)
    idx = np.empty(u.size, dtype=np.int64)
    cdf = np.empty(ll.size) if return_cdf else None
    st = lib().orc_resample_indices(
        ll.size, _p(ll), _p(lp), _p(lq), beta0, beta, int(uniform_weights), u.size, _p(u), _p(idx),
        _p(cdf) if return_cdf else None,
    )
    _check(st, beta)
    return (idx, cdf) if return_cdf else idx


def gather_rows(idx, x, ll, lp, lq):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    x = np.ascontiguousarray(x)
    assert x.dtype in (np.float64, np.float32)
    ll, lp, lq = _f64(ll), _f64(lp), _f64(lq)
    n_out, d = idx.size, x.shape[1]
    xo = np.empty((n_out, d), dtype=x.dtype)
    llo, lpo, lqo = np.empty(n_out), np.empty(n_out), np.empty(n_out)
    lib().orc_gather_rows(
        n_out, _p(idx), d, x.dtype.itemsize, _p(x), _p(xo), _p(ll), _p(lp), _p(lq), _p(llo), _p(lpo), _p(lqo)
    )
    return xo, llo, lpo, lqo


def systematic_uniforms(n_out, u0):
    u = np.empty(n_out)
    lib().orc_systematic_uniforms(n_out, u0, _p(u))
    return u


def stratified_uniforms(v):
    v = _f64(v)
    u = np.empty_like(v)
    lib().orc_stratified_uniforms(v.size, _p(v), _p(u))
    return u


def log_p_t(ll, lp, lq, beta):
    return lib().orc_log_p_t(ll, lp, lq, beta)


def compact_valid(x, ll, lp, lq):
    x, ll, lp, lq = _f64(x), _f64(ll), _f64(lp), _f64(lq)
    n, d = x.shape
    xo, llo, lpo, lqo = np.empty_like(x), np.empty(n), np.empty(n), np.empty(n)
    k = lib().orc_compact_valid(n, d, _p(x), _p(ll), _p(lp), _p(lq), _p(xo), _p(llo), _p(lpo), _p(lqo))
    return xo[:k], llo[:k], lpo[:k], lqo[:k]


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def pcn_noise(seed, gid, step, d, noise="f64"):
    xi = np.empty(d)
    u = ctypes.c_double(0)
    fn = lib().orc_pcn_noise if noise == "f64" else lib().orc_pcn_noise_f32
    fn(seed, gid, step, d, _p(xi), ctypes.addressof(u))
    return xi, u.value


class Mixture:
    """Diagonal Gaussian mixture log-density parameters (logw includes normalisation)."""

    def __init__(self, logw, mu, prec):
        self.logw = _f64(np.atleast_1d(logw))
        self.mu = _f64(np.atleast_2d(mu))
        self.prec = _f64(np.atleast_2d(prec))
        self.C, self.d = self.mu.shape
        assert self.prec.shape == self.mu.shape and self.logw.shape == (self.C,)

    def c_struct(self):
        return _Mixture(self.C, _p(self.logw), _p(self.mu), _p(self.prec))

    def logpdf(self, x):
        x = _f64(np.atleast_2d(x))
        return np.array(
            [
                lib().orc_diag_mixture_logpdf(self.d, self.C, _p(self.logw), _p(self.mu), _p(self.prec), _p(r))
                for r in x
            ]
        )


def pcn_step(x, ll, lp, lq, beta, mu, L, Linv, rho, t_ll, t_lp, t_lq, seed, gid0, step, noise="f64"):
    """In-place pCN step on numpy arrays; returns #accepted."""
    assert x.dtype == np.float64 and x.flags.c_contiguous
    n, d = x.shape
    mu, L, Linv = _f64(mu), _f64(L), _f64(Linv)
    a, b, c = t_ll.c_struct(), t_lp.c_struct(), t_lq.c_struct()
    return lib().orc_pcn_step(
        n, d, _p(x), _p(ll), _p(lp), _p(lq), beta, _p(mu), _p(L), _p(Linv), rho,
        ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c), seed, gid0, step, int(noise == "f32"),
    )


def gamma_unit(shape, seed, gid, step, noise="f64"):
    """Unit-scale Gamma(shape) variate of particle `gid` at Markov step `step` (tpCN scale mixture)."""
    return lib().orc_gamma_unit_mode(float(shape), seed, gid, step, int(noise == "f32"))


def tpcn_corr(q, d, nu):
    return np.array([lib().orc_tpcn_corr(float(v), int(d), float(nu)) for v in np.atleast_1d(q)])


def tpcn_step(x, ll, lp, lq, beta, mu, L, Linv, rho, nu, t_ll, t_lp, t_lq, seed, gid0, step, noise="f64"):
    """In-place t-preconditioned Crank-Nicolson step (Student-t reference with `nu` degrees of freedom)."""
    assert x.dtype == np.float64 and x.flags.c_contiguous
    n, d = x.shape
    mu, L, Linv = _f64(mu), _f64(L), _f64(Linv)
    a, b, c = t_ll.c_struct(), t_lp.c_struct(), t_lq.c_struct()
    return lib().orc_tpcn_step(
        n, d, _p(x), _p(ll), _p(lp), _p(lq), beta, _p(mu), _p(L), _p(Linv), rho, float(nu),
        ctypes.addressof(a), ctypes.addressof(b), ctypes.addressof(c), seed, gid0, step, int(noise == "f32"),
    )


def pcn_adapt(rho, acc, target, t):
    return lib().orc_pcn_adapt(rho, acc, target, t)


def moments(x):
    x = _f64(x)
    n, d = x.shape
    mean, cov = np.empty(d), np.empty((d, d))
    lib().orc_moments(n, d, _p(x), _p(mean), _p(cov))
    return mean, cov


def is_iteration(x, ll, lp, lq, beta0, target_eff, tol, rng_state):
    """One IS-only temperature iteration (bench cpu_baseline, kind "port")."""
    x, ll, lp, lq = _f64(x), _f64(ll), _f64(lp), _f64(lq)
    n, d = x.shape
    xo, llo, lpo, lqo = np.empty_like(x), np.empty(n), np.empty(n), np.empty(n)
    sc = np.zeros(6)
    st = lib().orc_is_iteration(
        n, d, _p(x), _p(ll), _p(lp), _p(lq), beta0, target_eff, tol, _p(rng_state),
        _p(xo), _p(llo), _p(lpo), _p(lqo), _p(sc),
    )
    _check(st)
    return (xo, llo, lpo, lqo), sc


def coupling_logprob(x, weights, biases, loc, scale):
    """fp32 log-density of the RealNVP coupling flow; weights/biases: 3 per coupling layer, torch Linear layout."""
    x = _f64(np.atleast_2d(x))
    n, d = x.shape
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    hidden = ws[0].shape[0]
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    out = np.empty(n)
    st = lib().orc_coupling_logprob(n, d, _p(x), len(ws) // 3, hidden, wp, bp, loc.ctypes.data, scale.ctypes.data, _p(out))
    if st != 0:
        raise ValueError(f"orc_coupling_logprob failed ({st})")
    return out


def maf_logprob(x, weights, biases, loc, scale, affine=0):
    """fp32 log-density of the masked autoregressive flow; weights: the MASKED matrices, 3 per transform, torch Linear layout.
    affine = 1: zuko's soft-clipped monotonic affine form (orc_maf_logprob_form)."""
    x = _f64(np.atleast_2d(x))
    n, d = x.shape
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    hidden = ws[0].shape[0]
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    out = np.empty(n)
    st = lib().orc_maf_logprob_form(n, d, _p(x), len(ws) // 3, hidden, wp, bp, loc.ctypes.data, scale.ctypes.data, int(affine), _p(out))
    if st != 0:
        raise ValueError(f"orc_maf_logprob failed ({st})")
    return out


class accept_margins:
    """`with accept_margins(n) as m:` - the pCN / tpCN / flow step functions called inside record particle i's accept margin
    log_a - log u in m[i] (the decision is m[i] > 0); used to show that mismatching decisions are razor edges."""

    def __init__(self, n):
        self.m = np.full(int(n), np.nan)

    def __enter__(self):
        lib().orc_set_margin_sink(_p(self.m))
        return self.m

    def __exit__(self, *exc):
        lib().orc_set_margin_sink(None)
        return False


def max_threads() -> int:
    return int(lib().orc_max_threads())


def pcn_flow_step(x, ll, lp, lq, beta, mu, L, Linv, rho, t_ll, t_lp, weights, biases, loc, scale, seed, gid0, step,
                  noise="f64", n_threads=1, flow_kind="coupling"):
    """In-place pCN step whose proposal density is a coupling flow (configs[2]); `n_threads` OpenMP threads
    (0 = every core of the host).  Returns #accepted."""
    assert x.dtype == np.float64 and x.flags.c_contiguous
    n, d = x.shape
    mu, L, Linv = _f64(mu), _f64(L), _f64(Linv)
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    a, b = t_ll.c_struct(), t_lp.c_struct()
    return lib().orc_pcn_flow_step_kind(n, d, _p(x), _p(ll), _p(lp), _p(lq), beta, _p(mu), _p(L), _p(Linv), rho,
                                        ctypes.addressof(a), ctypes.addressof(b), len(ws) // 3, ws[0].shape[0], wp, bp,
                                        loc.ctypes.data, scale.ctypes.data, seed, gid0, step, int(noise == "f32"), int(n_threads),
                                        {"coupling": 0, "maf": 1, "maf_softclip": 2}[flow_kind])


def tpcn_flow_step(x, ll, lp, lq, beta, mu, L, Linv, rho, nu, t_ll, t_lp, weights, biases, loc, scale, seed, gid0, step,
                   noise="f64", n_threads=1, flow_kind="coupling"):
    """In-place t-preconditioned Crank-Nicolson step (Student-t reference, `nu` > 0 degrees of freedom) whose proposal
    density is a neural flow: the reference's default pairing (smc/minipcn.py:46-49 `step_fn="tpcn"` around
    smc/base.py:507-519 with flows/torch/flows.py:140 `flow_class="MAF"`).  Returns #accepted."""
    assert x.dtype == np.float64 and x.flags.c_contiguous
    n, d = x.shape
    mu, L, Linv = _f64(mu), _f64(L), _f64(Linv)
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    a, b = t_ll.c_struct(), t_lp.c_struct()
    r = lib().orc_tpcn_flow_step_kind(n, d, _p(x), _p(ll), _p(lp), _p(lq), beta, _p(mu), _p(L), _p(Linv), rho, float(nu),
                                      ctypes.addressof(a), ctypes.addressof(b), len(ws) // 3, ws[0].shape[0], wp, bp,
                                      loc.ctypes.data, scale.ctypes.data, seed, gid0, step, int(noise == "f32"), int(n_threads),
                                      {"coupling": 0, "maf": 1, "maf_softclip": 2}[flow_kind])
    if r < 0:
        raise ValueError(f"orc_tpcn_flow_step_kind failed ({r})")
    return r


def transform(x, kind, periodic, lower, upper, mean=None, std=None, eps=1e-6, inverse=False):
    """CompositeTransform forward / inverse on rows of x: returns (y, log|det J|)."""
    x = _f64(np.atleast_2d(x))
    n, d = x.shape
    kind = np.ascontiguousarray(kind, dtype=np.int32)
    periodic = np.ascontiguousarray(periodic, dtype=np.int32)
    lower, upper = _f64(lower), _f64(upper)
    out, logj = np.empty_like(x), np.empty(n)
    m = None if mean is None else _f64(mean)
    s = None if std is None else _f64(std)
    st = lib().orc_transform(n, d, _p(x), _p(out), _p(logj), kind.ctypes.data, periodic.ctypes.data, _p(lower), _p(upper),
                             None if m is None else _p(m), None if s is None else _p(s), float(eps), int(bool(inverse)))
    if st != 0:
        raise ValueError(f"orc_transform failed ({st})")
    return out, logj
