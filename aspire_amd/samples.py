"""Particle containers with the reference's API surface (src/aspire/samples.py).

`BaseSamples` / `Samples` / `SMCSamples` keep the reference's field names, constructor arguments
and method names (samples.py:36-595, :1208-1332).  The difference is WHERE the per-particle
arithmetic runs: every SMC method (`log_weights`, `log_evidence_ratio`, `..._variance`, `resample`)
is executed by the engine — the HIP kernels of libasmc_hip.so — on device-resident tensors, and only
scalars come back to the host.  Without a HIP device these methods raise (no CPU fallback); tests
inject the oracle-backed test double through the `engine` field.
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass, field, fields
from typing import Any, Callable

import numpy as np
import torch

from . import smc_math
from ._xp import asarray, default_dtype, is_torch, is_torch_namespace, namespace_of, resolve_dtype, resolve_xp, to_numpy
from .comm import Comm

logger = logging.getLogger(__name__)

_default_engine = None


def get_default_engine():
    """Process-wide HipEngine on the current device (created on first use; raises without a GPU)."""
    global _default_engine
    if _default_engine is None:
        from .engine import HipEngine

        _default_engine = HipEngine(torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return _default_engine


def set_default_engine(engine) -> None:
    global _default_engine
    _default_engine = engine


@dataclass
class BaseSamples:
    """samples.py:36-413 (the parts the SMC path touches)."""

    x: Any
    log_likelihood: Any | None = None
    log_prior: Any | None = None
    log_q: Any | None = None
    parameters: list[str] | None = None
    dtype: Any | str | None = None
    xp: Callable | None = None
    device: Any = None

    def __post_init__(self):
        if self.xp is None:
            self.xp = namespace_of(self.x)
        if self.dtype is not None:
            self.dtype = resolve_dtype(self.dtype, self.xp)
        elif hasattr(self.x, "dtype") and self._is_float(self.x):
            self.dtype = resolve_dtype(self.x.dtype, self.xp)  # keep the precision the caller handed over
        else:
            self.dtype = default_dtype(self.xp)
        if self.device is None and is_torch(self.x):
            self.device = self.x.device
        self.x = self.array_to_namespace(self.x, dtype=self.dtype)
        if self.log_likelihood is not None:
            self.log_likelihood = self.array_to_namespace(self.log_likelihood, dtype=self._scalar_dtype())
        if self.log_prior is not None:
            self.log_prior = self.array_to_namespace(self.log_prior, dtype=self._scalar_dtype())
        if self.log_q is not None:
            self.log_q = self.array_to_namespace(self.log_q, dtype=self._scalar_dtype())
        if self.parameters is None:
            self.parameters = [f"x_{i}" for i in range(self.dims)]

    @staticmethod
    def _is_float(a) -> bool:
        dt = a.dtype
        return dt in (torch.float32, torch.float64) if isinstance(dt, torch.dtype) else np.issubdtype(dt, np.floating)

    def _scalar_dtype(self):
        """dtype of the per-particle scalars.  Device-resident (torch) state keeps log-probabilities
        in fp64 whatever the x dtype; host (numpy) containers follow the reference and
        use the sample dtype (samples.py:83-92)."""
        if is_torch_namespace(self.xp):
            return torch.float64
        return self.dtype

    @property
    def dims(self):
        if self.x is None:
            return 0
        return self.x.shape[1] if self.x.ndim > 1 else 1

    def array_to_namespace(self, x, dtype=None):
        return asarray(x, self.xp, dtype=dtype, device=self.device)

    def to_numpy(self, dtype: Any | str | None = None):
        return self.__class__(
            x=to_numpy(self.x),
            parameters=self.parameters,
            log_likelihood=None if self.log_likelihood is None else to_numpy(self.log_likelihood),
            log_prior=None if self.log_prior is None else to_numpy(self.log_prior),
            log_q=None if self.log_q is None else to_numpy(self.log_q),
            xp=np,
        )

    def to_namespace(self, xp, dtype: Any | str | None = None):
        return self.__class__(
            x=self.x,
            parameters=self.parameters,
            log_likelihood=self.log_likelihood,
            log_prior=self.log_prior,
            log_q=self.log_q,
            xp=xp,
            device=self.device if is_torch_namespace(xp) else None,
            dtype=dtype,
        )

    def to_dict(self, flat: bool = True, copy: bool = True):
        out = {}
        for f in fields(self):
            if f.name in ("x", "xp", "engine", "comm"):
                continue
            out[f.name] = getattr(self, f.name)
        out["xp"] = self.xp
        samples = dict(zip(self.parameters, to_numpy(self.x).T))
        if flat:
            out.update(samples)
        else:
            out["samples"] = samples
        return out

    @classmethod
    def from_dict(cls, dictionary):
        """samples.py:181-207: flat ({parameter: column, ...}) or nested ({"samples": {...}}) dictionaries."""
        dictionary = dict(dictionary)
        if "samples" in dictionary:
            samples = dictionary.pop("samples")
            parameters = dictionary.pop("parameters")
            if parameters is None:
                parameters = sorted(samples.keys())
            x = np.stack([np.asarray(samples[p]) for p in parameters], axis=-1)
        else:
            parameters = dictionary.pop("parameters")
            if parameters is None:
                raise ValueError("Parameters must be provided if samples are not nested in a 'samples' key")
            x = np.stack([np.asarray(dictionary[p]) for p in parameters], axis=-1)
            for p in parameters:
                dictionary.pop(p, None)
        known = {f.name for f in fields(cls) if f.init}
        return cls(x=x, parameters=parameters, **{k: v for k, v in dictionary.items() if k in known})

    def _encode_for_hdf5(self, flat: bool = True) -> dict:
        """samples.py:282-287."""
        d = self.to_numpy().to_dict(flat=flat)
        d.pop("device", None)
        # the namespace and the dtype travel as the reference writes them (utils.py:544-565): the module's name, and the dtype as
        # {"__dtype__": True, "xp": <module name>, "dtype": <name>} - `load` then hands the samples back in THAT namespace
        xp_name = getattr(self.xp, "__name__", "numpy")
        dt = self.dtype if self.dtype is not None else getattr(self.x, "dtype", None)
        dt_name = None if dt is None else (str(dt).split(".")[-1] if isinstance(dt, torch.dtype) else np.dtype(dt).name)
        d["dtype"] = None if dt_name is None else {"__dtype__": True, "xp": xp_name, "dtype": dt_name}
        d["xp"] = xp_name
        return d

    def save(self, h5_file, path: str = "samples", flat: bool = False):
        """samples.py:289-305: converted to numpy, flattened into datasets under the group `path`."""
        from .io import recursively_save_to_h5_file

        recursively_save_to_h5_file(h5_file, path, self._encode_for_hdf5(flat=flat))

    @classmethod
    def load(cls, h5_file, path: str = "samples"):
        """samples.py:307-322."""
        from .io import load_from_h5_file

        d = load_from_h5_file(h5_file, path)
        xp_name = d.get("xp")
        xp_name = xp_name if isinstance(xp_name, str) else "numpy"
        xp = resolve_xp(xp_name)  # ("numpy", "torch", or array_api_compat's wrappers of the two where that package is installed)
        enc = d.get("dtype")
        name = enc.get("dtype") if isinstance(enc, dict) else (enc if isinstance(enc, str) else None)
        d["xp"] = xp
        d["dtype"] = None if name is None else resolve_dtype(str(name).split(".")[-1], xp)
        return cls.from_dict(d)

    def to_dataframe(self, include: list | None = None):
        """samples.py:209-243: parameters as columns plus log_likelihood / log_prior / log_q (NaN when missing)."""
        import pandas as pd

        data = dict(zip(self.parameters, to_numpy(self.x).T))
        for key in (["log_likelihood", "log_prior", "log_q"] if include is None else include):
            v = getattr(self, key)
            data[key] = to_numpy(v) if v is not None else np.full(len(self.x), np.nan)
        return pd.DataFrame(data)

    def __getstate__(self):
        """Pickle as host arrays; device handles (engine, communicator, memoised reductions) are dropped."""
        state = dict(self.__dict__)
        for k in ("engine", "comm", "_wstats", "_ws1p", "_spec", "_spec_pending", "_moments"):
            state.pop(k, None)
        for k, v in list(state.items()):
            if is_torch(v):
                state[k] = to_numpy(v) if not is_torch_namespace(self.xp) else v.detach().cpu()
        if state.get("xp") is not None:
            state["xp"] = state["xp"].__name__  # modules do not pickle: the namespace travels by name
        return state

    def __setstate__(self, state):
        if isinstance(state.get("xp"), str):
            state["xp"] = resolve_xp(state["xp"])
        self.__dict__.update(state)
        self.__dict__.setdefault("engine", None)
        self.__dict__.setdefault("comm", None)

    def __str__(self):
        return f"No. samples: {len(self.x)}\nNo. parameters: {self.x.shape[-1]}\n"

    def __len__(self):
        return len(self.x)

    def __getitem__(self, idx):
        return self.__class__(
            x=self.x[idx],
            log_likelihood=self.log_likelihood[idx] if self.log_likelihood is not None else None,
            log_prior=self.log_prior[idx] if self.log_prior is not None else None,
            log_q=self.log_q[idx] if self.log_q is not None else None,
            parameters=self.parameters,
            dtype=self.dtype,
        )

    def __setitem__(self, idx, value):
        raise NotImplementedError("Setting items is not supported")

    @classmethod
    def concatenate(cls, samples: list["BaseSamples"]):
        if not samples:
            raise ValueError("No samples to concatenate")
        if not all(s.parameters == samples[0].parameters for s in samples):
            raise ValueError("Parameters do not match")
        if not all(s.xp == samples[0].xp for s in samples):
            raise ValueError("Array namespaces do not match")
        if not all(s.dtype == samples[0].dtype for s in samples):
            raise ValueError("Dtypes do not match")
        xp = samples[0].xp
        cat = (lambda arrs: torch.cat(arrs, dim=0)) if is_torch_namespace(xp) else (lambda arrs: np.concatenate(arrs, axis=0))

        def opt(name):
            vals = [getattr(s, name) for s in samples]
            return cat(vals) if all(v is not None for v in vals) else None

        return cls(x=cat([s.x for s in samples]), log_likelihood=opt("log_likelihood"), log_prior=opt("log_prior"),
                   log_q=opt("log_q"), parameters=samples[0].parameters, dtype=samples[0].dtype)

    @classmethod
    def from_samples(cls, samples: "BaseSamples", **kwargs):
        xp = kwargs.pop("xp", samples.xp)
        device = kwargs.pop("device", samples.device)
        kwargs.pop("dtype", None)  # reference computes but never forwards it (samples.py:378-392)
        return cls(x=samples.x, log_likelihood=samples.log_likelihood, log_prior=samples.log_prior,
                   log_q=samples.log_q, parameters=samples.parameters, xp=xp, device=device,
                   dtype=samples.dtype if is_torch_namespace(xp) == is_torch_namespace(samples.xp) else None, **kwargs)


@dataclass
class Samples(BaseSamples):
    """samples.py:416-595.  Importance weights are computed when all three log-terms are given."""

    log_w: Any = field(init=False)
    weights: Any = field(init=False)
    evidence: float = field(init=False)
    evidence_error: float = field(init=False)
    log_evidence: float | None = None
    log_evidence_error: float | None = None
    effective_sample_size: float = field(init=False)

    def __post_init__(self):
        super().__post_init__()
        if all(v is not None for v in (self.log_likelihood, self.log_prior, self.log_q)):
            self.compute_weights()
        else:
            self.log_w = None
            self.weights = None
            self.evidence = None
            self.evidence_error = None
            self.effective_sample_size = None

    @property
    def efficiency(self):
        if self.log_w is None:
            raise RuntimeError("Samples do not contain weights!")
        return self.effective_sample_size / len(self.x)

    def compute_weights(self):
        """samples.py:457-475 — one-shot importance weights (not the SMC hot path; plain array ops)."""
        xp = torch if is_torch_namespace(self.xp) else np

        def lse(v):
            c = v.max()
            return c + xp.log(xp.exp(v - c).sum())

        self.log_w = self.log_likelihood + self.log_prior - self.log_q
        self.log_evidence = lse(self.log_w) - math.log(len(self.x))
        self.weights = xp.exp(self.log_w)
        self.evidence = xp.exp(self.log_evidence)
        n = len(self.x)
        self.evidence_error = xp.sqrt(((self.weights - self.evidence) ** 2).sum() / (n * (n - 1)))
        self.log_evidence_error = abs(self.evidence_error / self.evidence)
        log_w = self.log_w - self.log_w.max()
        self.effective_sample_size = xp.exp(lse(log_w) * 2 - lse(log_w * 2))

    @property
    def scaled_weights(self):
        xp = torch if is_torch_namespace(self.xp) else np
        return xp.exp(self.log_w - self.log_w.max())

    def rejection_sample(self, rng=None):
        """samples.py:481-494: keep sample i with probability w_i / max w (returns x, log_likelihood, log_prior)."""
        if rng is None:
            rng = np.random.default_rng()
        log_u = np.log(rng.uniform(size=len(self.x)))
        log_w = to_numpy(self.log_w)
        accept = (log_w - log_w.max()) > log_u
        idx = accept if not is_torch(self.x) else torch.as_tensor(accept, device=self.x.device)
        return self.__class__(x=self.x[idx], log_likelihood=self.log_likelihood[idx], log_prior=self.log_prior[idx],
                              dtype=self.dtype, parameters=self.parameters, xp=self.xp)

    def __str__(self):
        out = super().__str__()
        if self.log_evidence is not None:
            out += f"Log evidence: {float(self.log_evidence):.2f} +/- {float(self.log_evidence_error):.2f}\n"
        if self.log_w is not None:
            out += f"Effective sample size: {float(self.effective_sample_size):.1f}\nEfficiency: {float(self.efficiency):.2f}\n"
        return out

    def to_namespace(self, xp):
        dev = self.device if is_torch_namespace(xp) else None
        conv = lambda v: None if v is None else asarray(v, xp, device=dev)  # noqa: E731
        return self.__class__(x=conv(self.x), parameters=self.parameters, log_likelihood=conv(self.log_likelihood),
                              log_prior=conv(self.log_prior), log_q=conv(self.log_q), xp=xp,
                              log_evidence=self.log_evidence, log_evidence_error=self.log_evidence_error)

    def to_numpy(self):
        conv = lambda v: None if v is None else to_numpy(v)  # noqa: E731
        return self.__class__(x=to_numpy(self.x), parameters=self.parameters, log_likelihood=conv(self.log_likelihood),
                              log_prior=conv(self.log_prior), log_q=conv(self.log_q),
                              log_evidence=self.log_evidence, log_evidence_error=self.log_evidence_error)

    def to_dataframe(self, include: list | None = None):
        """samples.py:559-578: the importance weights' logarithm next to the three log-probabilities by default."""
        return super().to_dataframe(["log_likelihood", "log_prior", "log_q", "log_w"] if include is None else include)

    def __getitem__(self, idx):
        sliced = super().__getitem__(idx)
        sliced.log_evidence = self.log_evidence
        sliced.log_evidence_error = self.log_evidence_error
        return sliced


@dataclass
class SMCSamples(BaseSamples):
    """samples.py:1208-1332 — tempered particle population; the arithmetic runs in the engine."""

    beta: float | None = None
    log_evidence: float | None = None
    log_evidence_error: float | None = None
    engine: Any = field(default=None, repr=False, compare=False)
    comm: Any = field(default=None, repr=False, compare=False)

    # ---- engine plumbing ------------------------------------------------------------------
    def _eng(self):
        if self.engine is None:
            self.engine = get_default_engine()
        return self.engine

    def _comm(self):
        if self.comm is None:
            self.comm = Comm()
        return self.comm

    def _dev3(self):
        """(ll, lp, lq) as contiguous fp64 tensors on the engine's device."""
        e = self._eng()
        return tuple(e.asarray(v) for v in (self.log_likelihood, self.log_prior, self.log_q))

    def _n_global(self) -> int:
        # sharded runs in owner layout have ragged shards: the population size travels with the object then
        return self.__dict__.get("n_global") or int(sum(self.shard_counts_list()))

    def _from_device(self, t):
        if is_torch_namespace(self.xp):
            return t
        return to_numpy(t)

    # ---- reference API --------------------------------------------------------------------
    def log_p_t(self, beta):
        """samples.py:1217-1219 (host-side convenience; the mutation kernel fuses this)."""
        log_p_T = self.log_likelihood + self.log_prior
        return (1 - beta) * self.log_q + beta * log_p_T

    def _stats(self, beta: float) -> smc_math.Stats:
        return self.weight_stats([float(beta)])[0]

    def weight_stats(self, betas) -> list:
        """(m, S1, S2) of the log-sum-exp of unnormalized_log_weights(beta) per candidate, memoised on this object
        (its log-probabilities never change): ESS, evidence ratio, variance and the resampling weights at one
        beta all come from the same device pass."""
        cache = self.__dict__.setdefault("_wstats", {})
        betas = [float(b) for b in betas]
        missing = [b for b in dict.fromkeys(betas) if b not in cache]
        if missing:
            ll, lp, lq = self._dev3()
            for b, st in zip(missing, smc_math.global_stats(self._eng(), self._comm(), ll, lp, lq, float(self.beta), missing,
                                                            self._n_global())):
                cache[b] = st
        return [cache[b] for b in betas]

    def remember_stats(self, beta: float, st, s1p: float | None = None):
        """Record results computed elsewhere (device-side beta search, evidence-variance pass)."""
        if st is not None:
            self.__dict__.setdefault("_wstats", {})[float(beta)] = st
        if s1p is not None:
            self.__dict__.setdefault("_ws1p", {})[float(beta)] = float(s1p)

    def speculate_importance_step(self, target_eff: float, tol: float, rng, *, resample_mode: str = "exact",
                                  resample_method: str = "multinomial", moments_n: int | None = None,
                                  defer: bool = False, shard_layout: str = "owner", factor_ahead: bool = False) -> bool:
        """Enqueue the whole importance step of one iteration - adaptive-beta search (smc/base.py:167-186), evidence
        moments (samples.py:1226-1242) and the multinomial resampling of all N particles at beta* (samples.py:1251-1287)
        - as one chain of launches with a single host synchronisation (include/asmc.h asmc_importance_step), and park
        the results on this object: the sampler's `determine_beta` then finds the search result, and `resample(beta*)`
        the resampled rows.  Speculative: the host's schedule rules (min / max beta step) may pick another beta than
        beta*, in which case `resample` ignores the parked rows and the generator has not been touched.
        `moments_n`: also enqueue the column sums and the centred Gram matrix of the resampled rows (centre = sums /
        moments_n; `engine.mean_gram`) behind the gather, so that the mutation's reference fit costs no pass and no
        synchronisation of its own; they travel with the resampled population (`_moments`).
        `factor_ahead` (with `moments_n`): the reference Gaussian's factorisation (`engine.reference_factor`: covariance ->
        Cholesky factor -> inverse, one block) goes behind the moments as well - it depends on the resampled rows only, not on
        what the host decides from the step's scalars, so it runs while the host is still reading them.
        `defer`: enqueue only; the caller has more to wait for on the stream and calls `finish_speculation()` itself.
        Sharded populations (owner layout): the chain runs up to the ranks' offspring counts (`smc_math.shard_step_enqueue`:
        search, moments, weights, global cdf slice, draw selection, with the collectives on the stream between them);
        `finish_speculation` synchronises once and puts the search and the gather behind it.
        Returns False when the step does not apply (non-PCG64 generator, other resampling schemes, slot layout, engines
        without these entry points)."""
        self.__dict__.pop("_spec", None)
        self.__dict__.pop("_spec_pending", None)
        e, comm = self._eng(), self._comm()
        st4 = smc_math.pcg64_state(rng)
        if st4 is None or resample_mode != "exact" or resample_method != "multinomial" or not float(self.beta) < 1.0:
            return False
        if comm.sharded:
            if shard_layout != "owner" or not smc_math.shard_step_available(e, comm):
                return False
            ll, lp, lq = self._dev3()
            x = e.asarray(self.x, dtype=self.x.dtype if is_torch(self.x) else torch.float64)
            n = self._n_global()
            h = smc_math.shard_step_enqueue(e, comm, ll, lp, lq, float(self.beta), float(target_eff), float(tol), n,
                                            self.shard_counts_list(), st4, n)
            self._spec_pending = dict(key=(float(target_eff), float(tol)), shard=h, src=(x, ll, lp, lq), rng=rng,
                                      state=[int(v) for v in st4], n=n, moments_n=int(moments_n) if moments_n else None,
                                      factor_ahead=bool(factor_ahead))
            if not defer:
                self.finish_speculation()
            return True
        if not hasattr(e, "importance_step"):
            return False
        ll, lp, lq = self._dev3()
        x = e.asarray(self.x, dtype=self.x.dtype if is_torch(self.x) else torch.float64)
        n = ll.numel()
        if getattr(e, "importance_step_disabled", False):
            return False
        idx = e.importance_step(ll, lp, lq, float(self.beta), float(target_eff), float(tol), st4, n)
        rows = e.gather(idx, x, ll, lp, lq)
        if hasattr(e, "importance_result_enqueue"):
            e.importance_result_enqueue()  # the step's scalars come back as soon as the step is done, not behind the moments
        with_moments = (bool(moments_n) and hasattr(e, "mean_gram_enqueue")
                        and e.mean_gram_enqueue(rows[0], int(moments_n), gathered=True))  # (the rows come straight from the gather)
        factor = None
        if with_moments and factor_ahead and hasattr(e, "reference_factor"):
            factor = e.reference_factor(int(rows[0].shape[1]), int(moments_n), int(moments_n))
            factor = (*factor, getattr(e, "ref_generation", None))  # (mu, L, Linv, the request's generation: its status cell)
        self._spec_pending = dict(key=(float(target_eff), float(tol)), rows=rows, rng=rng, state=[int(v) for v in st4], n=n,
                                  moments_n=int(moments_n) if with_moments else None, gram_gen=getattr(e, "_gram_gen", None),
                                  factor=factor)
        if not defer:
            self.finish_speculation()
        return True

    def finish_speculation(self):
        """Wait for an enqueued importance step and park its results (`speculate_importance_step`)."""
        p = self.__dict__.pop("_spec_pending", None)
        if p is None:
            return
        e = self._eng()
        if "shard" in p:
            comm = self._comm()
            h = p["shard"]
            rows, moments = None, None
            if hasattr(e, "shard_step_finish"):
                # the wait, and right behind it - from C, no interpreter in between - this rank's search and gather (samples.py:
                # 1278-1287: its sub-sequence of Generator.choice's index vector, then its rows) into buffers sized before the
                # wait: shares stay within 1/world (1 +- SHARD_IMBALANCE) or the step is redone phase by phase anyway
                cap = min(h["n_out"], int((1.0 + 2.0 * smc_math.SHARD_IMBALANCE) * h["n_out"] / h["world"]) + 2048)
                search, parts, info, rows = e.shard_step_finish(h["res"], h["world"], h["rank"], h["cdf"], h["buf"], cap, *p["src"],
                                                                rec_token=h.get("rec_token", 0))
            else:
                search, parts, info, u_kept = smc_math.shard_step_wait(e, comm, h)
                if u_kept is not None:
                    idx = e.search(h["cdf"], u_kept)
                    if hasattr(e, "rec_claim"):
                        e.rec_claim(h.get("rec_token", 0), *p["src"][1:])
                    rows = e.gather(idx, *p["src"])
            if rows is not None:
                # (enqueued BEFORE the host has examined the gathered numbers - a step it then rejects (rare) just drops the rows)
                if (p["moments_n"] is not None and hasattr(e, "mean_gram_enqueue")
                        and e.mean_gram_enqueue(rows[0], p["moments_n"], comm, gathered=True)):
                    factor = None
                    if p.get("factor_ahead") and hasattr(e, "reference_factor"):
                        factor = e.reference_factor(int(rows[0].shape[1]), p["moments_n"], p["moments_n"])
                        factor = (*factor, getattr(e, "ref_generation", None))
                    moments = (rows[0].data_ptr(), tuple(rows[0].shape), p["moments_n"], e._gram_gen, factor)
            ok, m2, _, new_counts = smc_math.shard_step_check(p["shard"], search, parts, info)
            ok = ok and rows is not None
            if not ok:
                rows, moments = None, None
            self._spec = dict(key=p["key"], search=search, found=bool(ok), beta=float(search[0]), rows=rows, m2=m2, rng=p["rng"],
                              state=p["state"], n=p["n"], moments=moments, counts=new_counts)
            return
        b, eff1, conv, passes, n_nan, trip, trip_one, m2, _, found = e.importance_result()
        moments, rows = None, p["rows"]
        if p["moments_n"] is not None:  # still on the stream: fetched by the reference fit (HipSMC._fit_reference_gaussian)
            moments = (rows[0].data_ptr(), tuple(rows[0].shape), p["moments_n"], p["gram_gen"], p.get("factor"))
        self._spec = dict(key=p["key"], search=(b, eff1, conv, passes, n_nan, trip, trip_one),
                          found=bool(found and conv), beta=float(b), rows=rows, m2=m2, rng=p["rng"],
                          state=p["state"], n=p["n"], moments=moments)

    def _take_speculated(self, beta: float, n_samples: int, rng, resample_mode: str, resample_method: str):
        """The rows `speculate_importance_step` parked, if they are the answer to this `resample` call; else None."""
        spec = self.__dict__.pop("_spec", None)
        if (spec is None or not spec["found"] or spec["beta"] != float(beta) or spec["n"] != n_samples
                or spec["rng"] is not rng or resample_mode != "exact" or resample_method != "multinomial"):
            return None
        st4 = smc_math.pcg64_state(rng)
        if st4 is None or [int(v) for v in st4] != spec["state"]:
            return None
        return spec

    def unnormalized_log_weights(self, beta: float):
        """samples.py:1221-1224."""
        ll, lp, lq = self._dev3()
        return self._from_device(self._eng().log_weights(ll, lp, lq, float(self.beta), float(beta), 0.0))

    def log_evidence_ratio(self, beta: float) -> float:
        """samples.py:1226-1228."""
        return smc_math.log_evidence_ratio(self._stats(beta))

    def log_evidence_ratio_variance(self, beta: float) -> float:
        """samples.py:1230-1242."""
        ll, lp, lq = self._dev3()
        st = self._stats(beta)
        return smc_math.evidence_variance(self._eng(), self._comm(), ll, lp, lq, float(self.beta), float(beta), st)

    def log_weights(self, beta: float):
        """samples.py:1244-1249 (raises ValueError on NaN log-weights)."""
        ll, lp, lq = self._dev3()
        st = self._stats(beta)
        shift = smc_math.log_evidence_ratio(st)
        return self._from_device(self._eng().log_weights(ll, lp, lq, float(self.beta), float(beta), shift))

    def effective_sample_size(self, beta: float) -> float:
        """utils.py:510-512 of log_weights(beta) without materialising the weights."""
        return smc_math.ess(self._stats(beta))

    def resample(self, beta, n_samples: int | None = None, rng: np.random.Generator = None, *,
                 resample_mode: str = "exact", resample_method: str = "multinomial", shard_layout: str = "owner",
                 want_variance: bool = False, moments_n: int | None = None):
        """samples.py:1251-1287.  `resample_mode`: "exact" (sequential-order cdf == numpy cumsum,
        bit-exact indices) or "fast"; `resample_method`: "multinomial" (reference) or the opt-in
        "systematic" / "stratified".  Sharded populations (`comm.sharded`): `shard_layout="owner"` keeps every
        offspring on its ancestor's rank (no row exchange; ragged shards), `"slots"` gives output slot j to rank
        j // n_local and reproduces the single-rank particle order (smc_math.resample_owner / resample_indices).
        `want_variance=True` also returns log_evidence_ratio_variance(beta), which the owner layout computes in the
        same exchange."""
        if rng is None:
            rng = np.random.default_rng()
        comm = self._comm()
        n_global = self._n_global()
        if n_samples is None:
            n_samples = n_global
        uniform = False
        if beta == self.beta:
            if n_samples is None or n_samples == n_global:
                logger.warning("Resampling with the same beta value, returning identical samples")
                return (self, 0.0) if want_variance else self
            uniform = True
        e = self._eng()
        ll, lp, lq = self._dev3()
        x = e.asarray(self.x, dtype=self.x.dtype if is_torch(self.x) else torch.float64)
        st = None if uniform else self._stats(beta)
        var = None

        def wrap(xo, llo, lpo, lqo, counts=None):
            out = self.__class__(x=self._from_device(xo), log_likelihood=self._from_device(llo),
                                 log_prior=self._from_device(lpo), log_q=self._from_device(lqo), beta=beta,
                                 dtype=self.dtype, parameters=self.parameters, xp=self.xp, engine=self.engine,
                                 comm=self.comm)
            # `moments_n`: the caller's mutation fits its reference Gaussian to the moments of these rows - start them now
            # (engine.mean_gram_enqueue; summed over the ranks of a sharded run), they are fetched when the fit needs them
            if (moments_n and spec is None and is_torch(xo) and hasattr(e, "mean_gram_enqueue")
                    and e.mean_gram_enqueue(xo, int(moments_n), comm, gathered=True)):  # (xo comes straight from a gather)
                out.__dict__["_moments"] = (xo.data_ptr(), tuple(xo.shape), int(moments_n), e._gram_gen)
            if comm.sharded:
                out.n_global = int(n_samples)
                out.shard_counts = [int(c) for c in counts]
                out.ragged = len(set(out.shard_counts)) > 1
            return (out, var) if want_variance else out

        counts = self.shard_counts_list()
        ragged = len(set(counts)) > 1

        spec = None if uniform else self._take_speculated(beta, int(n_samples), rng, resample_mode, resample_method)
        if spec is not None:
            # samples.py:1230-1242 from the sum the fused step left (smc_math.evidence_variance_and_lse's formula)
            mean_u = st.S1 / st.n
            var_u = spec["m2"] / st.n
            var = float(var_u / (st.n * (mean_u**2))) if mean_u != 0 else float("nan")
            rng.bit_generator.advance(int(n_samples))  # the n draws Generator.choice takes
            res = wrap(*spec["rows"], counts=spec.get("counts") or counts)
            if spec.get("moments") is not None:
                (res[0] if want_variance else res).__dict__["_moments"] = spec["moments"]
            return res

        if shard_layout == "owner" and smc_math.owner_layout_ok(e, comm, rng, resample_method, uniform):
            idx, var, s1p, new_counts = smc_math.resample_owner(
                e, comm, ll, lp, lq, float(self.beta), float(beta), int(n_samples), rng, mode=resample_mode, st=st,
                counts=counts, method=resample_method, uniform_weights=uniform)
            if idx is not None:
                return wrap(*e.gather(idx, x, ll, lp, lq), counts=new_counts)
            self.remember_stats(beta, None, s1p)  # shares too uneven: fall through to the slot layout, which rebalances
        if want_variance and var is None and not uniform:
            var, s1p = smc_math.evidence_variance_and_lse(e, comm, ll, lp, lq, float(self.beta), float(beta), st)
            self.remember_stats(beta, None, s1p)
        if ragged:  # the slot layout addresses particles as rank * n_local + i: equalise the shards first
            x, ll, lp, lq = rebalance_shards(e, comm, x, ll, lp, lq)
        idx, _ = smc_math.resample_indices(e, comm, ll, lp, lq, float(self.beta), float(beta), int(n_samples), rng,
                                           mode=resample_mode, method=resample_method, uniform_weights=uniform,
                                           st=st, s1p=self.__dict__.get("_ws1p", {}).get(float(beta)))
        per = -(-int(n_samples) // comm.world)  # output slots per rank (smc_math.resample_indices)
        slot_counts = [max(0, min(per, int(n_samples) - r * per)) for r in range(comm.world)]
        return wrap(*gather_global(e, comm, idx, x, ll, lp, lq), counts=slot_counts)

    def shard_counts_list(self) -> list:
        """Rows per rank of this (possibly sharded) population, in rank order."""
        comm = self._comm()
        c = self.__dict__.get("shard_counts")
        if c is not None and len(c) == comm.world:
            return [int(v) for v in c]
        return [len(self.x)] * comm.world

    def gid0(self) -> int:
        """Global index of this rank's first particle (keys the per-particle noise streams)."""
        comm = self._comm()
        return int(sum(self.shard_counts_list()[:comm.rank]))

    def __str__(self):
        out = super().__str__()
        if self.log_evidence is not None:
            out += f"Log evidence: {float(self.log_evidence):.2f}\n"
        return out

    def to_standard_samples(self):
        """samples.py:1295-1305 — drops log_q on purpose, keeps the evidence."""
        return Samples(x=self.x, log_likelihood=self.log_likelihood, log_prior=self.log_prior, xp=self.xp,
                       parameters=self.parameters, log_evidence=self.log_evidence,
                       log_evidence_error=self.log_evidence_error)

    def to_numpy(self):
        conv = lambda v: None if v is None else to_numpy(v)  # noqa: E731
        return self.__class__(x=to_numpy(self.x), parameters=self.parameters, log_likelihood=conv(self.log_likelihood),
                              log_prior=conv(self.log_prior), log_q=conv(self.log_q), beta=self.beta,
                              log_evidence=self.log_evidence, log_evidence_error=self.log_evidence_error,
                              engine=self.engine, comm=self.comm)

    def __getitem__(self, idx):
        sliced = super().__getitem__(idx)
        sliced.beta = self.beta
        sliced.log_evidence = self.log_evidence
        sliced.log_evidence_error = self.log_evidence_error
        sliced.engine, sliced.comm = self.engine, self.comm
        return sliced


def rebalance_shards(engine, comm, x, ll, lp, lq):
    """Equal shards again (N / world rows per rank) without changing the global (rank-major) particle order: every
    rank ships the row ranges that fall into another rank's slice with one variable all-to-all."""
    world, rank, n_loc = comm.world, comm.rank, x.shape[0]
    counts = comm.all_gather_f64(np.array([float(n_loc)])).reshape(-1).astype(np.int64)
    total = int(counts.sum())
    if total % world:
        raise ValueError(f"cannot split {total} particles evenly over {world} ranks")
    per = total // world
    starts = np.concatenate([[0], np.cumsum(counts)])

    def overlap(a0, a1, b0, b1):
        return int(max(0, min(a1, b1) - max(a0, b0)))

    send = [overlap(starts[rank], starts[rank + 1], r * per, (r + 1) * per) for r in range(world)]
    recv = [overlap(starts[s], starts[s + 1], rank * per, (rank + 1) * per) for s in range(world)]
    packed = torch.stack([ll, lp, lq], dim=1).contiguous()
    x2 = comm.all_to_all_rows(x.contiguous(), send, recv)
    s2 = comm.all_to_all_rows(packed, send, recv)
    return x2, s2[:, 0].contiguous(), s2[:, 1].contiguous(), s2[:, 2].contiguous()


def gather_global(engine, comm, idx, x, ll, lp, lq):
    """Rows `idx` (GLOBAL indices) of the sharded population -> this rank's new shard.

    World 1: one row-gather kernel (samples.py:1279-1287).  Sharded: requests are bucketed by owner
    rank, exchanged with one all-to-all of the index lists, served by the owner's gather kernel and
    returned with one all-to-all of rows; a final local gather restores the output order."""
    if not comm.sharded:
        return engine.gather(idx, x, ll, lp, lq)
    n_local = x.shape[0]
    d = x.shape[1]
    owner = torch.div(idx, n_local, rounding_mode="floor")
    order = torch.argsort(owner, stable=True)
    send_idx = (idx - owner * n_local)[order].contiguous()
    counts = torch.bincount(owner, minlength=comm.world).cpu().tolist()
    all_counts = comm.all_gather_f64(np.array(counts, dtype=np.float64)).astype(np.int64)  # [world(src), world(dst owner)]
    recv_counts = all_counts[:, comm.rank].tolist()  # how many rows each rank asks of me
    req = comm.all_to_all_rows(send_idx, counts, recv_counts)
    if req.numel() > 0:
        rx, rll, rlp, rlq = engine.gather(req.contiguous(), x, ll, lp, lq)
    else:
        rx = x[:0]
        rll = rlp = rlq = ll[:0]
    packed = torch.empty((rx.shape[0], 3), dtype=torch.float64, device=rx.device)
    if rx.shape[0] > 0:
        packed[:, 0], packed[:, 1], packed[:, 2] = rll, rlp, rlq
    got_x = comm.all_to_all_rows(rx.contiguous(), recv_counts, counts)
    got_s = comm.all_to_all_rows(packed, recv_counts, counts)
    # rows arrive grouped by owner in `order`; undo the permutation with a local gather
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=order.device)
    return engine.gather(inv.contiguous(), got_x.contiguous(), got_s[:, 0].contiguous(), got_s[:, 1].contiguous(),
                         got_s[:, 2].contiguous())
