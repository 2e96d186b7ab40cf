"""Sampler base classes (reference src/aspire/samplers/base.py:19-286, samplers/mcmc.py:14-110).

The constructor contract is the reference's `aspire.samplers` entry-point contract
(samplers/base.py:40-50) so the classes here could be registered into a real aspire install.
"""
from __future__ import annotations

import logging
import pickle
from pathlib import Path
from typing import Any, Callable

import numpy as np
import torch

from .._xp import is_torch, is_torch_namespace, to_numpy
from ..comm import Comm, default_comm
from ..samples import Samples, get_default_engine
from ..targets import DiagGaussianMixture

logger = logging.getLogger(__name__)


def track_calls(func):
    """utils.py:1003-1029: remember the positional and keyword arguments of every call of a sampler's `sample` method
    (`config_dict` reports them; the checkpoint state carries that config).  The log lives on the instance, keyed by the
    method's qualified name, so a subclass's `sample` and the base `sample` it calls keep separate logs as in the
    reference."""
    import functools

    @functools.wraps(func)
    def wrapper(self, *args, **kwargs):
        log = self.__dict__.setdefault("_call_log", {}).setdefault(func.__qualname__, {"args": [], "kwargs": []})
        log["args"].append(args)
        log["kwargs"].append(kwargs)
        return func(self, *args, **kwargs)

    return wrapper


class IdentityTransform:
    """transforms.py:125-139 — the default "smc" preconditioning for unbounded problems
    (aspire.py:337-341: affine_transform=False, bounded_to_unbounded=False)."""

    def __init__(self, xp=None, dtype=None):
        self.xp, self.dtype = xp, dtype

    def fit(self, x):
        return x

    def forward(self, x):
        return x, None

    def inverse(self, z):
        return z, None

    def new_instance(self, xp=None):
        return IdentityTransform(xp=xp or self.xp, dtype=self.dtype)

    def config_dict(self):
        return {}


class Sampler:
    """samplers/base.py:19-91."""

    def __init__(self, log_likelihood: Callable, log_prior: Callable, dims: int, prior_flow, xp: Callable,
                 dtype: Any | str | None = None, parameters: list[str] | None = None,
                 preconditioning_transform: Callable | None = None, engine=None, comm: Comm | None = None):
        self.prior_flow = prior_flow
        self._log_likelihood = log_likelihood
        self._log_prior = log_prior
        self.dims = dims
        self.xp = xp if xp is not None else np
        self.backend_str = "torch" if is_torch_namespace(self.xp) else "numpy"
        self.dtype = dtype
        self.parameters = parameters
        self.history = None
        self.n_likelihood_evaluations = 0
        self._last_checkpoint_state: dict | None = None
        self._last_checkpoint_bytes: bytes | None = None
        self.preconditioning_transform = preconditioning_transform or IdentityTransform(xp=self.xp)
        self._engine = engine
        self._comm = comm

    # ---- engine / communicator ----------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_default_engine()
        return self._engine

    @property
    def comm(self) -> Comm:
        if self._comm is None:
            self._comm = default_comm(getattr(self.engine, "device", "cpu"))
        return self._comm

    @property
    def x_torch_dtype(self):
        if self.dtype is None:
            return torch.float64
        if isinstance(self.dtype, torch.dtype):
            return self.dtype
        return {"float32": torch.float32, "float64": torch.float64}[np.dtype(self.dtype).name]

    def fit_preconditioning_transform(self, x):
        return self.preconditioning_transform.fit(x)

    def sample(self, n_samples: int) -> Samples:
        raise NotImplementedError

    # ---- user callables -------------------------------------------------------------------
    def _user_view(self, x_dev: torch.Tensor, **known):
        """The `samples` object handed to user callables: arrays in the sampler's namespace."""
        if is_torch_namespace(self.xp):
            return Samples(x_dev, xp=torch, parameters=self.parameters, **known)
        conv = {k: (None if v is None else to_numpy(v)) for k, v in known.items()}
        return Samples(to_numpy(x_dev), xp=np, parameters=self.parameters, **conv)

    def _to_dev(self, v) -> torch.Tensor:
        return self.engine.asarray(v if is_torch(v) else np.asarray(v, dtype=np.float64))

    def log_likelihood(self, samples) -> Any:
        """samplers/base.py:81-87 (counts evaluations)."""
        self.n_likelihood_evaluations += len(samples)
        return self._log_likelihood(samples)

    def log_prior(self, samples) -> Any:
        return self._log_prior(samples)

    def _eval_prior_likelihood(self, x_dev: torch.Tensor, log_q_dev: torch.Tensor | None = None):
        """log_prior then log_likelihood at x (prior first and stored on the samples object passed to
        the likelihood, as reference smc/base.py:512-513 / docs/recipes.rst rely on)."""
        n = x_dev.shape[0]
        if isinstance(self._log_prior, DiagGaussianMixture):
            lp = self.engine.mixture_logpdf(x_dev, self._log_prior.device_mixture(self.engine))
            view = None
        else:
            view = self._user_view(x_dev, log_q=log_q_dev)
            lp = self._to_dev(self.log_prior(view))
        if isinstance(self._log_likelihood, DiagGaussianMixture):
            self.n_likelihood_evaluations += n
            ll = self.engine.mixture_logpdf(x_dev, self._log_likelihood.device_mixture(self.engine))
        else:
            if view is None:
                view = self._user_view(x_dev, log_q=log_q_dev)
            view.log_prior = view.array_to_namespace(lp if is_torch_namespace(view.xp) else to_numpy(lp))
            ll = self._to_dev(self.log_likelihood(view))
        return lp, ll

    def _flow_log_prob(self, x_dev: torch.Tensor) -> torch.Tensor:
        """prior_flow.log_prob(x) (smc/base.py:510); host (numpy-namespace) flows get a host copy."""
        dev_flow = None
        if hasattr(self.prior_flow, "device_coupling") and hasattr(self.engine, "coupling_logprob"):
            try:  # float32 coupling flow of a supported shape: fp32 MFMA kernel instead of the torch modules
                dev_flow = self.prior_flow.device_coupling(self.engine)
            except (ValueError, RuntimeError):  # unsupported dtype / shape: the flow's own torch modules evaluate it
                dev_flow = None
        if dev_flow is not None:
            return self.engine.coupling_logprob(x_dev if x_dev.is_contiguous() else x_dev.contiguous(), dev_flow)
        flow_xp = getattr(self.prior_flow, "xp", None)
        arg = x_dev if (flow_xp is None or is_torch_namespace(flow_xp)) else to_numpy(x_dev)
        return self._to_dev(self.prior_flow.log_prob(arg))

    # ---- config / checkpoint plumbing (samplers/base.py:93-276, minimal) -------------------
    def config_dict(self, include_sample_calls: str | bool = "last") -> dict:
        """samplers/base.py:93-141."""
        config = {"sampler_class": self.__class__.__name__}
        if include_sample_calls is not False:
            if include_sample_calls is True:
                include_sample_calls = "all"
            if not isinstance(include_sample_calls, str):
                raise ValueError("include_sample_calls must be a string ('last' or 'all') or False."
                                 f"Received: {include_sample_calls} of type {type(include_sample_calls)}")
            calls = self.__dict__.get("_call_log", {}).get(getattr(type(self).sample, "__qualname__", ""))
            if calls is None:
                return config
            if include_sample_calls.lower() == "last":
                config["sample_calls"] = {"args": calls["args"][-1] if calls["args"] else None,
                                          "kwargs": calls["kwargs"][-1] if calls["kwargs"] else None}
            elif include_sample_calls.lower() == "all":
                config["sample_calls"] = {"args": {str(i): v for i, v in enumerate(calls["args"])},
                                          "kwargs": {str(i): v for i, v in enumerate(calls["kwargs"])}}
            else:
                raise ValueError("Invalid value for include_sample_calls. Must be 'last', 'all', or False.")
        return config

    def _checkpoint_extra_state(self) -> dict:
        """samplers/base.py:150-152."""
        return {}

    def _restore_extra_state(self, state: dict) -> None:
        """samplers/base.py:154-156."""
        _ = state

    def build_checkpoint_state(self, samples, iteration: int | None = None, meta: dict | None = None,
                               include_sample_calls: str | bool = "last") -> dict:
        """samplers/base.py:158-178: same keys, same order."""
        state = {"sampler": self.__class__.__name__, "iteration": iteration, "samples": samples,
                 "config": self.config_dict(include_sample_calls=include_sample_calls),
                 "parameters": self.parameters, "meta": meta or {}}
        state.update(self._checkpoint_extra_state())
        return state

    def serialize_checkpoint(self, state: dict, protocol: int | None = None) -> bytes:
        """samplers/base.py:180-187."""
        return pickle.dumps(state, protocol=pickle.HIGHEST_PROTOCOL if protocol is None else int(protocol))

    def default_checkpoint_callback(self, state: dict) -> None:
        """samplers/base.py:189-192: keep the latest checkpoint (state + pickled bytes) on the sampler."""
        self._last_checkpoint_state = state
        self._last_checkpoint_bytes = self.serialize_checkpoint(state)

    def _rank_path(self, file_path) -> Path:
        """Sharded runs: every rank checkpoints its own shard into its own file (`name.rank<r>.ext`)."""
        file_path = Path(file_path)
        if not self.comm.sharded:
            return file_path
        return file_path.with_name(f"{file_path.stem}.rank{self.comm.rank}{file_path.suffix}")

    def default_file_checkpoint_callback(self, file_path: str | Path | None):
        """samplers/base.py:194-216: overwrite `/checkpoint/state` of an HDF5 file (needs h5py); additionally a `.pkl`
        path writes the same pickled bytes to a plain file (h5py is not part of every image)."""
        if file_path is None:
            return self.default_checkpoint_callback
        file_path = self._rank_path(file_path)
        lower = file_path.name.lower()
        if not lower.endswith((".h5", ".hdf5", ".pkl", ".pickle")):
            raise ValueError("Checkpoint file must be an HDF5 file (.h5 or .hdf5) or a pickle file (.pkl).")

        def _callback(state: dict) -> None:
            if lower.endswith((".h5", ".hdf5")):
                from ..io import open_h5

                with open_h5(file_path, "a") as h5_file:
                    self.save_checkpoint_to_hdf(state, h5_file, path="checkpoint", dsetname="state")
                self.default_checkpoint_callback(state)
            else:
                self.default_checkpoint_callback(state)
                with open(file_path, "wb") as fp:
                    fp.write(self._last_checkpoint_bytes)

        return _callback

    def save_checkpoint_to_hdf(self, state: dict, h5_file, path: str = "sampler_checkpoints", dsetname: str | None = None,
                               protocol: int | None = None) -> None:
        """samplers/base.py:218-236: the state as ONE pickled `S1` byte dataset `<path>/<dsetname>`; `h5_file` is an open
        h5py.File or anything with its group protocol."""
        from ..io import dump_state

        if dsetname is None:
            dsetname = f"iter_{state.get('iteration', 'unknown')}"
        dump_state(state, h5_file, path=path, dsetname=dsetname, protocol=protocol or pickle.HIGHEST_PROTOCOL)

    def load_checkpoint_from_file(self, file_path: str | Path, h5_path: str = "checkpoint", dsetname: str = "state") -> dict:
        """samplers/base.py:238-253."""
        file_path = Path(file_path)
        if file_path.name.lower().endswith((".h5", ".hdf5")):
            from ..io import load_state, open_h5

            with open_h5(file_path, "r") as h5_file:
                return load_state(h5_file, h5_path, dsetname)
        with open(file_path, "rb") as f:
            return pickle.loads(f.read())

    def restore_from_checkpoint(self, source):
        """samplers/base.py:255-276."""
        if isinstance(source, (str, Path)):
            p = Path(source)
            state = self.load_checkpoint_from_file(p if p.exists() else self._rank_path(p))
        elif isinstance(source, (bytes, bytearray)):
            state = pickle.loads(source)
        elif isinstance(source, dict):
            state = source
        else:
            raise TypeError("Unsupported checkpoint source type.")
        samples_saved = state.get("samples")
        if samples_saved is None:
            raise ValueError("Checkpoint missing samples.")
        self._restore_extra_state(state)
        return samples_saved, state

    @property
    def last_checkpoint_state(self):
        return self._last_checkpoint_state

    @property
    def last_checkpoint_bytes(self):
        return self._last_checkpoint_bytes


class MCMCSampler(Sampler):
    """samplers/mcmc.py:14-110."""

    def __init__(self, log_likelihood, log_prior, dims, prior_flow, xp, dtype=None, parameters=None,
                 preconditioning_transform=None, rng=None, engine=None, comm=None):
        super().__init__(log_likelihood, log_prior, dims, prior_flow, xp, dtype, parameters,
                         preconditioning_transform, engine=engine, comm=comm)
        self.rng = rng or np.random.default_rng()

    def draw_initial_samples(self, n_samples: int) -> Samples:
        """samplers/mcmc.py:49-110: draw from the proposal flow until n rows have finite log-prior and
        log-likelihood; a non-finite log_q is an error.  State stays on device; the finite-mask
        compaction is the asmc_compact_valid kernel."""
        e = self.engine
        n_drawn = 0
        parts = []
        while n_drawn < n_samples:
            x, log_q = self.prior_flow.sample_and_log_prob(n_samples)
            x = e.asarray(x, dtype=self.x_torch_dtype)
            log_q = self._to_dev(log_q)
            n_nan, n_inf = e.count_nonfinite(log_q)
            if n_nan or n_inf:
                raise ValueError(
                    "Proposal returned non-finite log probabilities. "
                    "aspire assumes the proposal is a valid, normalized "
                    "probability distribution and should therefore only "
                    "return samples with finite log probabilities.")
            lp, ll = self._eval_prior_likelihood(x, log_q)
            xv, llv, lpv, lqv = e.compact_valid(x, ll, lp, log_q)
            n_valid = xv.shape[0]
            if n_valid < x.shape[0]:
                logger.debug("Proposal returned %d invalid samples with non-finite log prior or log "
                             "likelihood. These samples will be discarded.", x.shape[0] - n_valid)
            if n_valid > 0:
                parts.append((xv, llv, lpv, lqv))
                n_drawn += n_valid
        if len(parts) == 1:
            xv, llv, lpv, lqv = parts[0]
        else:
            xv, llv, lpv, lqv = (torch.cat([p[i] for p in parts], dim=0) for i in range(4))
        if n_drawn > n_samples:
            xv, llv, lpv, lqv = (t[:n_samples].contiguous() for t in (xv, llv, lpv, lqv))
        out = Samples(x=xv, xp=torch, parameters=self.parameters)
        # set after construction (as the reference does, mcmc.py:82-87) so no IS weights are formed
        out.log_likelihood, out.log_prior, out.log_q = llv, lpv, lqv
        return out
