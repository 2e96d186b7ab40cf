"""`Aspire` façade with the reference's API surface for the SMC path (src/aspire/aspire.py:25-570).

Kept: constructor keywords (aspire.py:79-98), `fit`, `sample_posterior` (signature, kwargs routing by
`inspect.signature`, `return_history`, `n_likelihood_evaluations`), `get_sampler_class`,
`init_sampler`.  Only `sampler in {"smc", "minipcn_smc"}` is implemented natively (the hot path this
repository replaces); other names resolve through the `aspire.samplers` entry-point group exactly
like the reference (aspire.py:293-304) and otherwise raise.  HDF5 checkpoint files, plotting and
the JAX backend are out of scope.
"""
from __future__ import annotations

import copy
import logging
from inspect import signature
from typing import Any, Callable

import numpy as np
import torch

from ._xp import is_torch_namespace, to_numpy
from .flows import CouplingFlow, Flow, GaussianFlow, MAFFlow
from .history import FlowHistory, History
from .samples import Samples
from .samplers.base import IdentityTransform, Sampler

logger = logging.getLogger(__name__)


class Aspire:
    """Accelerated Sequential Posterior Inference via REuse — SMC path on MI355X."""

    def __init__(self, *, log_likelihood: Callable, log_prior: Callable, dims: int,
                 parameters: list[str] | None = None, periodic_parameters: list[str] | None = None,
                 prior_bounds: dict[str, tuple[float, float]] | None = None, bounded_to_unbounded: bool = True,
                 bounded_transform: str = "logit", device: str | None = None, xp: Callable | None = None,
                 flow: Flow | None = None, flow_backend: str = "zuko", flow_matching: bool = False,
                 eps: float = 1e-6, dtype: Any | str | None = None, **kwargs) -> None:
        self.log_likelihood = log_likelihood
        self.log_prior = log_prior
        self.dims = dims
        self.parameters = parameters
        self.device = device
        self.eps = eps
        self.periodic_parameters = periodic_parameters
        self.prior_bounds = prior_bounds
        self.bounded_to_unbounded = bounded_to_unbounded
        self.bounded_transform = bounded_transform
        self.flow_matching = flow_matching
        self.flow_backend = flow_backend
        self.flow_kwargs = kwargs
        self.xp = xp
        self.dtype = dtype
        self._flow = flow
        self._sampler = None
        if flow_matching:
            raise NotImplementedError("flow matching is outside the SMC hot path (SURVEY.md §2)")

    @property
    def flow(self):
        return self._flow

    @flow.setter
    def flow(self, flow: Flow):
        self._flow = flow

    @property
    def sampler(self) -> Sampler | None:
        return self._sampler

    @property
    def n_likelihood_evaluations(self):
        return None if self._sampler is None else self._sampler.n_likelihood_evaluations

    def convert_to_samples(self, x, log_likelihood=None, log_prior=None, log_q=None, evaluate: bool = True, xp=None):
        """aspire.py:142-175."""
        samples = Samples(x=x, parameters=self.parameters, log_likelihood=log_likelihood, log_prior=log_prior,
                          log_q=log_q, xp=xp or self.xp, dtype=self.dtype)
        if evaluate:
            if log_prior is None:
                samples.log_prior = samples.array_to_namespace(self.log_prior(samples))
            if log_likelihood is None:
                samples.log_likelihood = samples.array_to_namespace(self.log_likelihood(samples))
            if samples.log_q is not None:
                samples.compute_weights()
        return samples

    def init_flow(self):
        """aspire.py:177-206.  Backends: "coupling" (PyTorch RealNVP, HIP kernels in the hot path), "maf" (masked autoregressive
        flow, the reference's default flow class; PyTorch passes only) and "gaussian" (analytic, HIP).
        "zuko" is accepted when zuko is importable and otherwise maps to "coupling" with a warning
        (zuko is not part of this image)."""
        backend = self.flow_backend.lower()
        # aspire.py:182-191: the flow works behind a FlowTransform.  The built-in flows standardise internally (CouplingFlow
        # loc/scale, GaussianFlow moments), so the transform is only materialised when it does more than an affine map:
        # finite prior bounds with bounded_to_unbounded.  Without it the fused device paths (log q inside the pCN kernels)
        # stay available.
        data_transform = None
        if self.prior_bounds is not None and self.bounded_to_unbounded:
            from .transforms import FlowTransform

            params = self.parameters if self.parameters is not None else [f"x_{i}" for i in range(self.dims)]
            data_transform = FlowTransform(parameters=params, prior_bounds=self.prior_bounds,
                                           bounded_to_unbounded=True, bounded_transform=self.bounded_transform,
                                           device=self.device, xp=self.xp, eps=self.eps, dtype=self.dtype)
            if data_transform.is_identity or not data_transform._kind.any():
                data_transform = None  # all bounds infinite: nothing beyond the internal standardisation
        if backend == "zuko":
            # the reference's default backend (aspire.py:79-98) with its default flow class, a masked autoregressive flow
            # (flows/torch/flows.py:140-164).  zuko itself is absent: `MAFFlow` restates the architecture and runs on the HIP kernels.
            # Other zuko classes (NSF, ...) have no counterpart: the coupling flow stands in, with a warning.
            fc = str(self.flow_kwargs.get("flow_class", "MAF")).upper()
            if fc != "MAF":
                logger.warning("zuko flow class %r is not available here; using the built-in coupling flow", fc)
                self.flow_kwargs = {k: v for k, v in self.flow_kwargs.items() if k != "flow_class"}
            backend = "maf" if fc == "MAF" else "coupling"
        if backend == "maf" or str(self.flow_kwargs.get("flow_class", "")).upper() == "MAF":
            # the reference's default flow class (ZukoFlow(flow_class="MAF"), flows/torch/flows.py:140-164): PyTorch passes only
            kw = {k: v for k, v in self.flow_kwargs.items() if k != "flow_class"}
            if "transforms" in kw:  # zuko's name for the number of autoregressive transforms
                kw["n_transforms"] = int(kw.pop("transforms"))
            known = {"n_transforms", "hidden_features", "seed", "flow_dtype"}
            unknown = sorted(k for k in kw if k not in known)
            if unknown:  # zuko-only options (passes, randperm, activation, ...): this flow has no counterpart - say so, go on
                logger.warning("MAFFlow ignores flow_kwargs %s (zuko options without a counterpart here)", unknown)
                kw = {k: v for k, v in kw.items() if k in known}
            self._flow = MAFFlow(dims=self.dims, device=self.device or "cpu", data_transform=data_transform,
                                 dtype=kw.pop("flow_dtype", torch.float32), **kw)
        elif backend == "coupling":
            kw = {k: v for k, v in self.flow_kwargs.items() if k != "flow_class"}
            if "transforms" in kw:  # zuko's name for the number of transforms
                kw["n_layers"] = int(kw.pop("transforms"))
            known = {"n_layers", "hidden_features", "seed", "flow_dtype"}
            unknown = sorted(k for k in kw if k not in known)
            if unknown:  # zuko-only options (bins, passes, randperm, ...) of the class this flow stands in for
                logger.warning("CouplingFlow ignores flow_kwargs %s (zuko options without a counterpart here)", unknown)
                kw = {k: v for k, v in kw.items() if k in known}
            self._flow = CouplingFlow(dims=self.dims, device=self.device or "cpu", data_transform=data_transform,
                                      dtype=kw.pop("flow_dtype", torch.float32), **kw)
        elif backend == "gaussian":
            self._flow = GaussianFlow(dims=self.dims, data_transform=data_transform, **self.flow_kwargs)
        else:
            raise ValueError(f"Unknown flow backend: {self.flow_backend}")

    def fit(self, samples: Samples, checkpoint_path: str | None = None, checkpoint_save_config: bool = True,
            overwrite: bool = False, **kwargs) -> History:
        """aspire.py:208-270 (flow training runs in PyTorch, upstream of the hot path)."""
        if self.xp is None:
            self.xp = samples.xp
        if self.parameters is None and samples.parameters is not None:
            self.parameters = samples.parameters.copy()
        if self.flow is None:
            self.init_flow()
        elif getattr(self, "_skip_flow_training", False) and not overwrite:  # aspire.py:239-243: a checkpointed flow was loaded
            logger.info("Skipping flow training because a checkpointed flow was loaded.")
            return FlowHistory()
        self.training_samples = samples
        logger.info(f"Training with {len(samples.x)} samples")
        history = self.flow.fit(samples.x, **kwargs) or FlowHistory()
        defaults = getattr(self, "_checkpoint_defaults", None)  # aspire.py:248-254: inside `auto_checkpoint`
        if checkpoint_path is None and defaults:  # (the config group is written once per context)
            checkpoint_path = defaults["path"]
            checkpoint_save_config = bool(defaults["save_config"]) and not defaults.get("saved_config", False)
            defaults["saved_config"] = defaults.get("saved_config", False) or bool(defaults["save_config"])
        # aspire.py:251-269: config group (once) and the flow (if missing, or overwrite).  Sharded runs: the file's shared groups
        # are written by rank 0 alone (HDF5 has one writer; the per-rank sampler state goes to `<stem>.rank<r>.<ext>`), the
        # other ranks wait behind a barrier so that nobody opens the file while it is being written
        comm = self._checkpoint_comm()
        if checkpoint_path is not None and comm.rank != 0:
            comm.barrier()
        elif checkpoint_path is not None:
            try:
                self._fit_checkpoint(checkpoint_path, checkpoint_save_config, overwrite)
            finally:
                comm.barrier()
        return history

    def _checkpoint_comm(self):
        """The communicator whose rank 0 owns the shared groups of a checkpoint file: the sampler's, else the process group's."""
        comm = getattr(self._sampler, "comm", None) if self._sampler is not None else None
        if comm is None:
            from .comm import default_comm

            comm = default_comm(self.device or "cpu")
        return comm

    def _fit_checkpoint(self, checkpoint_path, checkpoint_save_config: bool, overwrite: bool) -> None:
        """aspire.py:251-269 (rank 0 only in a sharded run)."""
        store = self._open_checkpoint_store(checkpoint_path)
        if store is None:
            self._write_config_sidecar(checkpoint_path, sampler=False, save_config=checkpoint_save_config)
            return
        with store as h5_file:
            if checkpoint_save_config:
                if "aspire_config" in h5_file:
                    del h5_file["aspire_config"]
                self.save_config(h5_file, include_sampler_config=False)
            if "flow" in h5_file:
                if overwrite:
                    del h5_file["flow"]
                    self._try_save_flow(h5_file)
            else:
                self._try_save_flow(h5_file)

    def _hdf5_route(self, path) -> bool:
        """Whether `checkpoint_path` is served as an HDF5 file (its name says so and h5py imports) - decided WITHOUT opening it:
        in a sharded run only rank 0 ever opens the shared file."""
        if not self._is_hdf5_path(path):
            return False
        from . import io

        if not io.h5py_available():
            logger.warning("HDF5 files need h5py, which is not installed; writing pickle checkpoints next to %s instead", path)
            return False
        return True

    # ---- checkpoint files (aspire.py:501-557, 799-870) ----------------------------------------------------------------
    @staticmethod
    def _is_hdf5_path(path) -> bool:
        return str(path).lower().endswith((".h5", ".hdf5"))

    def _open_checkpoint_store(self, path):
        """The file behind `checkpoint_path`, opened for appending: an h5py File (through `io.open_h5`, the reference's
        `AspireFile`) when the path names an HDF5 file and h5py imports, else None - the caller then takes the pickle route
        (`.pkl` sampler checkpoints + a JSON sidecar with the two config dictionaries)."""
        if not self._is_hdf5_path(path):
            return None
        from . import io

        try:
            return io.open_h5(path, "a")
        except RuntimeError as exc:  # h5py is not installed
            logger.warning("%s; writing pickle checkpoints next to %s instead", exc, path)
            return None

    @staticmethod
    def _pickle_checkpoint_path(path) -> str:
        from pathlib import Path

        p = Path(path)
        return str(p if p.name.lower().endswith((".pkl", ".pickle")) else p.with_suffix(".pkl"))

    def _write_config_sidecar(self, path, sampler: bool, save_config: bool = True) -> None:
        """Pickle route: `<checkpoint>.config.json` holds what the HDF5 route stores as the `aspire_config` and
        `sampler_config` groups."""
        if not save_config:
            return
        import json
        from pathlib import Path

        side = Path(self._pickle_checkpoint_path(path)).with_suffix(".config.json")
        doc = {"aspire_config": self.config_dict(include_sampler_config=False)}
        if sampler and self.sampler is not None:
            cfg = self.sampler.config_dict(include_sample_calls="last")
            if hasattr(self, "_last_sampler_type"):
                cfg["sampler_type"] = self._last_sampler_type
            doc["sampler_config"] = cfg
        with open(side, "w") as f:
            json.dump(doc, f, indent=2, default=str)

    def _try_save_flow(self, h5_file, path: str = "flow") -> bool:
        """`save_flow` for flows that can be stored (a flow with a data transform, or a user-supplied proposal without `save`,
        is skipped with a warning: the checkpoint stays usable, the flow has to be rebuilt by the caller)."""
        try:
            self.save_flow(h5_file, path=path)
            return True
        except (NotImplementedError, AttributeError) as exc:
            logger.warning("flow not saved to the checkpoint file: %s", exc)
            if path in h5_file:
                del h5_file[path]
            return False

    def save_config(self, h5_file, path: str = "aspire_config", **kwargs) -> None:
        """aspire.py:799-819: the configuration as one group of dotted datasets."""
        from .io import recursively_save_to_h5_file

        recursively_save_to_h5_file(h5_file, path, self.config_dict(**kwargs))

    def save_sampler_config(self, h5_file, path: str = "sampler_config", **kwargs) -> None:
        """aspire.py:821-851: the configuration of the last sampler (+ `sampler_type`)."""
        from .io import recursively_save_to_h5_file

        config = self.sampler.config_dict(**kwargs) if self.sampler else {}
        if hasattr(self, "_last_sampler_type"):
            config["sampler_type"] = self._last_sampler_type
        recursively_save_to_h5_file(h5_file, path, config)

    def save_flow(self, h5_file, path: str = "flow") -> None:
        """aspire.py:853-866."""
        if self.flow is None:
            raise ValueError("Flow has not been initialized.")
        self.flow.save(h5_file, path=path)

    def load_flow(self, h5_file, path: str = "flow") -> None:
        """aspire.py:868-883: the flow class follows `flow_backend` / `flow_kwargs["flow_class"]` as in `init_flow`."""
        backend = str(self.flow_backend).lower()
        fc = str(self.flow_kwargs.get("flow_class", "MAF")).upper()
        if backend == "gaussian":
            raise ValueError("the analytic Gaussian proposal has no parameters to load")
        cls = MAFFlow if (backend == "maf" or (backend == "zuko" and fc == "MAF") or fc == "MAF" and backend != "coupling") else CouplingFlow
        self._flow = cls.load(h5_file, path=path, device=self.device or "cpu")

    def get_sampler_class(self, sampler_type: str) -> Callable:
        """aspire.py:272-305."""
        if sampler_type == "importance":
            from .samplers.importance import ImportanceSampler as SamplerClass
        elif sampler_type in ["smc", "minipcn_smc"]:
            from .samplers.smc import HipSMC as SamplerClass
        else:
            from importlib.metadata import entry_points

            eps = {ep.name: ep for ep in entry_points(group="aspire.samplers")}
            if sampler_type in eps:
                SamplerClass = eps[sampler_type].load()
            else:
                raise ValueError(f"Unknown sampler type: {sampler_type}")
        return SamplerClass

    def init_sampler(self, sampler_type: str, preconditioning: str | None = None,
                     preconditioning_kwargs: dict | None = None, **kwargs) -> Callable:
        """aspire.py:307-381.  "default" preconditioning for smc is affine_transform=False,
        bounded_to_unbounded=False (aspire.py:337-341): the identity for unbounded problems."""
        SamplerClass = self.get_sampler_class(sampler_type)
        if sampler_type != "importance" and preconditioning is None:
            preconditioning = "default"
        preconditioning = preconditioning.lower() if preconditioning else None
        if preconditioning is None or preconditioning == "none":
            transform = None
        elif preconditioning in ["standard", "default"]:
            pk = dict(preconditioning_kwargs or {})
            pk.setdefault("affine_transform", False)  # aspire.py:337-341
            pk.setdefault("bounded_to_unbounded", False)
            pk.setdefault("bounded_transform", "logit")
            needs_bounds = pk["bounded_to_unbounded"] and self.prior_bounds is not None
            if pk["affine_transform"] or needs_bounds or self.periodic_parameters:
                from .transforms import CompositeTransform

                params = self.parameters if self.parameters is not None else [f"x_{i}" for i in range(self.dims)]
                transform = CompositeTransform(parameters=params, prior_bounds=self.prior_bounds,
                                               periodic_parameters=self.periodic_parameters, xp=self.xp,
                                               device=self.device, dtype=self.dtype, **pk)
            else:  # the reference builds a CompositeTransform with every stage off: the identity
                transform = IdentityTransform(xp=self.xp)
        elif preconditioning == "flow":  # aspire.py:351-366: a flow refitted to the particles at every temperature
            from .transforms import FlowPreconditioningTransform

            pk = dict(preconditioning_kwargs or {})
            pk.setdefault("affine_transform", False)
            params = self.parameters if self.parameters is not None else [f"x_{i}" for i in range(self.dims)]
            transform = FlowPreconditioningTransform(parameters=params, flow_backend=self.flow_backend,
                                                     flow_kwargs=self.flow_kwargs, flow_matching=self.flow_matching,
                                                     periodic_parameters=self.periodic_parameters,
                                                     bounded_to_unbounded=self.bounded_to_unbounded, prior_bounds=self.prior_bounds,
                                                     xp=self.xp, dtype=self.dtype, device=self.device, **pk)
        else:
            raise ValueError(f"Unknown preconditioning: {preconditioning}")
        return SamplerClass(log_likelihood=self.log_likelihood, log_prior=self.log_prior, dims=self.dims,
                            prior_flow=self.flow, xp=self.xp, dtype=self.dtype,
                            preconditioning_transform=transform, parameters=self.parameters, **kwargs)

    def _sample_checkpoint_config(self, checkpoint_path, hdf5_route: bool, save_config: bool, saved_flow: bool) -> None:
        """aspire.py:539-557 (rank 0 only in a sharded run)."""
        if not hdf5_route:
            self._write_config_sidecar(checkpoint_path, sampler=True, save_config=save_config)
            return
        with self._open_checkpoint_store(checkpoint_path) as h5_file:
            if save_config:
                for grp in ("aspire_config", "sampler_config"):
                    if grp in h5_file:
                        del h5_file[grp]
                self.save_config(h5_file, include_sampler_config=False)
                self.save_sampler_config(h5_file, include_sample_calls="last")
            if self.flow is not None and not saved_flow and "flow" not in h5_file:
                self._try_save_flow(h5_file)

    def sample_posterior(self, n_samples: int | None = None, sampler: str = "importance", xp: Any = None,
                         return_history: bool = False, preconditioning: str | None = None,
                         preconditioning_kwargs: dict | None = None, checkpoint_path: str | None = None,
                         checkpoint_every: int = 1, checkpoint_save_config: bool = True, **kwargs) -> Samples:
        """aspire.py:383-570."""
        # aspire.py:453-467: after `resume_from_file` / inside `auto_checkpoint(resume=True)` the call continues the saved run
        if sampler == "importance" and getattr(self, "_resume_sampler_type", None):
            sampler = self._resume_sampler_type
        if "resume_from" not in kwargs and hasattr(self, "_resume_from_default"):
            kwargs["resume_from"] = self._resume_from_default
            kwargs.update(getattr(self, "_resume_overrides", {}) or {})
            if n_samples in (None, 1000) and getattr(self, "_resume_n_samples", None):
                n_samples = self._resume_n_samples
        defaults = getattr(self, "_checkpoint_defaults", None)  # aspire.py:490-494
        if checkpoint_path is None and defaults:
            checkpoint_path, checkpoint_every, checkpoint_save_config = defaults["path"], defaults["every"], defaults["save_config"]
        SamplerClass = self.get_sampler_class(sampler)
        init_params = signature(SamplerClass.__init__).parameters
        sampler_kwargs = {k: v for k, v in kwargs.items() if k in init_params and k != "self"}
        kwargs = {k: v for k, v in kwargs.items() if k not in init_params or k == "self"}
        self._sampler = self.init_sampler(sampler, preconditioning=preconditioning,
                                          preconditioning_kwargs=preconditioning_kwargs, **sampler_kwargs)
        self._last_sampler_type = sampler
        saved_flow, hdf5_route = False, False
        comm = self._checkpoint_comm()
        if checkpoint_path is not None:  # aspire.py:501-529
            supports = {"checkpoint_file_path", "checkpoint_every"}.issubset(signature(self._sampler.sample).parameters)
            hdf5_route = self._hdf5_route(checkpoint_path)
            if not supports:
                logger.warning(f"Sampler {sampler} does not support checkpointing. Checkpoint will not be saved.")
            else:
                kwargs.setdefault("checkpoint_file_path", checkpoint_path if hdf5_route else self._pickle_checkpoint_path(checkpoint_path))
                kwargs.setdefault("checkpoint_every", checkpoint_every)
            # sharded runs: the file's shared groups (flow, aspire_config, sampler_config) and the JSON sidecar are written by rank 0
            # alone, the others wait behind a barrier; the sampler's own state is per rank (`<stem>.rank<r>.<ext>`, samplers/base.py)
            try:
                if hdf5_route and comm.rank == 0:
                    with self._open_checkpoint_store(checkpoint_path) as h5_file:
                        if self.flow is not None and "flow" not in h5_file:
                            saved_flow = self._try_save_flow(h5_file)
            finally:
                comm.barrier()
        samples = self._sampler.sample(n_samples, **kwargs)
        if checkpoint_path is not None:  # aspire.py:539-557: the two config groups behind the sampler's own /checkpoint/state
            try:
                if comm.rank == 0:
                    self._sample_checkpoint_config(checkpoint_path, hdf5_route, checkpoint_save_config, saved_flow)
            finally:
                comm.barrier()
        self._last_sample_posterior_kwargs = {
            "n_samples": n_samples, "sampler": sampler, "xp": xp, "return_history": return_history,
            "preconditioning": preconditioning, "preconditioning_kwargs": preconditioning_kwargs,
            "sampler_init_kwargs": sampler_kwargs,
            "sample_kwargs": {k: v for k, v in kwargs.items() if k not in ("rng", "checkpoint_callback")},
            "checkpoint_path": None if checkpoint_path is None else str(checkpoint_path),
        }
        out_xp = xp if xp is not None else (self.xp if self.xp is not None else np)
        samples = samples.to_namespace(out_xp) if is_torch_namespace(out_xp) else samples.to_numpy()
        samples.parameters = self.parameters if self.parameters is not None else samples.parameters
        logger.info(f"Sampled {len(samples)} samples from the posterior")
        logger.info(f"Number of likelihood evaluations: {self.n_likelihood_evaluations}")
        if return_history:
            return samples, self._sampler.history
        return samples

    # ---- resuming a saved run (aspire.py:573-760, 911-1200) -----------------------------------------------------------------
    @staticmethod
    def _plain(value):
        """What an HDF5 / JSON config leaf becomes as a constructor argument."""
        if isinstance(value, np.generic):
            return value.item()
        if isinstance(value, np.ndarray):
            return value.tolist()
        if isinstance(value, dict):
            return {k: Aspire._plain(v) for k, v in value.items()}
        if isinstance(value, (list, tuple)):
            return [Aspire._plain(v) for v in value]
        return value

    @classmethod
    def _load_resume_data(cls, file_path, checkpoint_path: str = "checkpoint", checkpoint_dset: str = "state",
                          config_path: str = "aspire_config", sampler_config_path: str = "sampler_config") -> dict:
        """aspire.py:911-999: what a checkpoint file says about the run it belongs to - the two config dictionaries, the sampler
        type, the number of samples the run asked for and whether a sampler state is there.  HDF5 files through `io.open_h5`;
        without h5py (or for a `.pkl` path) the pickle route's `<stem>.pkl` + `<stem>.config.json`."""
        from pathlib import Path

        from . import io

        out = {"aspire_config": None, "sampler_config": None, "sampler_type": None, "n_samples": None, "state": None,
               "state_path": None, "hdf5": False}
        path = Path(file_path)
        if cls._is_hdf5_path(path) and io.h5py_available() and path.is_file():
            out["hdf5"] = True
            with io.open_h5(path, "r") as h5_file:
                if config_path in h5_file:
                    out["aspire_config"] = cls._plain(io.load_from_h5_file(h5_file, config_path))
                if sampler_config_path in h5_file:
                    out["sampler_config"] = cls._plain(io.load_from_h5_file(h5_file, sampler_config_path))
                try:
                    out["state"] = io.load_state(h5_file, checkpoint_path, checkpoint_dset)
                    out["state_path"] = str(path)
                except Exception:
                    logger.warning("Checkpoint not found at %s/%s in %s; will resume without a checkpoint.", checkpoint_path,
                                   checkpoint_dset, path)
        else:
            import json
            import pickle

            pkl = Path(cls._pickle_checkpoint_path(path))
            side = pkl.with_suffix(".config.json")
            if side.is_file():
                doc = json.load(open(side))
                out["aspire_config"], out["sampler_config"] = doc.get("aspire_config"), doc.get("sampler_config")
            rank_pkl = pkl if pkl.is_file() else None
            if rank_pkl is not None:
                try:
                    out["state"] = pickle.loads(rank_pkl.read_bytes())
                    out["state_path"] = str(pkl)
                except Exception:
                    logger.warning("Failed to decode checkpoint; proceeding without resume state.")
            elif any(pkl.parent.glob(f"{pkl.stem}.rank*{pkl.suffix}")):
                out["state_path"] = str(pkl)  # a sharded run: every rank restores its own `<stem>.rank<r>.pkl` (samplers/base.py)
        sc = out["sampler_config"]
        if isinstance(sc, dict):
            out["sampler_type"] = sc.get("sampler_type")
            out["n_samples"] = cls._resume_n_samples_from_sampler_config(sc)
        if out["n_samples"] is None and isinstance(out["state"], dict) and out["state"].get("samples") is not None:
            out["n_samples"] = len(out["state"]["samples"])
        return out

    @staticmethod
    def _resume_n_samples_from_sampler_config(sampler_config) -> int | None:
        """aspire.py:1030-1063: the `n_samples` of the saved `sample()` call."""
        calls = sampler_config.get("sample_calls") if isinstance(sampler_config, dict) else None
        if not isinstance(calls, dict):
            return None
        args = calls.get("args")
        try:
            if args is not None and not isinstance(args, (str, bytes, dict)) and len(args) > 0:
                return int(args[0])
            kw = calls.get("kwargs")
            if isinstance(kw, dict) and "n_samples" in kw:
                return int(kw["n_samples"])
        except (TypeError, ValueError):
            return None
        return None

    def _set_resume_defaults(self, data: dict, sampler: str | None = None, resume_kwargs: dict | None = None) -> None:
        """aspire.py:1001-1028: the next `sample_posterior` call continues from the saved state (nothing to do without one)."""
        if data.get("state_path") is None:
            return
        state = data.get("state") or {}
        self._resume_from_default = data["state_path"]
        self._resume_sampler_type = sampler or data.get("sampler_type") or (state.get("sampler") if isinstance(state, dict) else None)
        if self._resume_sampler_type == "HipSMC":  # (a state records the class name, `sample_posterior` takes the registry name)
            self._resume_sampler_type = "smc"
        self._resume_n_samples = data.get("n_samples")
        self._resume_overrides = dict(resume_kwargs or {})
        self._resume_sampler_config = {k: v for k, v in (data.get("sampler_config") or {}).items() if k != "sampler_class"}

    @classmethod
    def resume_from_file(cls, file_path: str, *, log_likelihood: Callable, log_prior: Callable, sampler: str | None = None,
                         checkpoint_path: str = "checkpoint", checkpoint_dset: str = "state", flow_path: str = "flow",
                         config_path: str = "aspire_config", resume_kwargs: dict | None = None, **aspire_kwargs):
        """aspire.py:573-646: rebuild the object from ONE file (config + flow + sampler state) and arm it so that the next
        `sample_posterior()` continues the saved run.  The two densities are not stored and must be passed.  Extra keywords
        (`engine=...`, `flow=...` for a proposal the file could not store) go to the constructor."""
        from . import io

        data = cls._load_resume_data(file_path, checkpoint_path, checkpoint_dset, config_path)
        cfg = data["aspire_config"]
        if cfg is None:
            raise ValueError(f"Config path '{config_path}' not found in {file_path}")
        cfg = {k: v for k, v in cfg.items() if k not in ("sampler_config", "sampler_type", "log_likelihood", "log_prior")}
        xp_name = cfg.pop("xp", None)
        flow_kwargs = cfg.pop("flow_kwargs", None) or {}
        if isinstance(cfg.get("prior_bounds"), dict):
            cfg["prior_bounds"] = {k: tuple(v) for k, v in cfg["prior_bounds"].items()}
        xp = None
        if isinstance(xp_name, str):
            xp = np if "numpy" in xp_name else (torch if "torch" in xp_name else None)
        given_flow = aspire_kwargs.pop("flow", None)
        aspire = cls(log_likelihood=log_likelihood, log_prior=log_prior, xp=xp, flow=given_flow,
                     **{**cfg, **(flow_kwargs if isinstance(flow_kwargs, dict) else {}), **aspire_kwargs})
        if given_flow is None:
            loaded = False
            if data["hdf5"]:
                with io.open_h5(file_path, "r") as h5_file:
                    if flow_path in h5_file:
                        aspire.load_flow(h5_file, path=flow_path)
                        loaded = True
            if not loaded:
                raise ValueError(f"Flow path '{flow_path}' not found in {file_path} (pass flow=... for a proposal the file does not hold)")
        aspire._set_resume_defaults(data, sampler=sampler, resume_kwargs=resume_kwargs)
        aspire._skip_flow_training = True
        aspire._checkpoint_defaults = {"path": file_path, "every": 1, "save_config": False, "save_flow": False,
                                       "saved_config": False, "saved_flow": False}
        return aspire

    def auto_checkpoint(self, path: str, every: int = 1, save_config: bool = True, save_flow: bool = True, resume: bool = False):
        """aspire.py:648-747: inside the context `fit` / `sample_posterior` write to `path` by default; with `resume=True` and an
        existing file the saved flow is loaded (training is then skipped) and the next `sample_posterior()` continues the run."""
        from contextlib import contextmanager

        @contextmanager
        def ctx():
            attrs = ["_checkpoint_defaults", "_resume_from_default", "_resume_sampler_type", "_resume_n_samples", "_resume_overrides",
                     "_resume_sampler_config", "_skip_flow_training"]
            saved = {a: getattr(self, a) for a in attrs if hasattr(self, a)}
            self._checkpoint_defaults = {"path": path, "every": every, "save_config": save_config, "save_flow": save_flow,
                                         "saved_config": False, "saved_flow": False}
            if resume:
                from . import io

                data = self._load_resume_data(path)
                if data["state_path"] is not None or data["aspire_config"] is not None:
                    logger.info(f"Resuming from checkpoint file at {path}")
                    if data["hdf5"]:
                        try:
                            with io.open_h5(path, "r") as h5_file:
                                if "flow" in h5_file:
                                    self.load_flow(h5_file, path="flow")
                        except (ValueError, KeyError) as exc:
                            logger.warning("Flow not loaded from %s: %s", path, exc)
                    self._set_resume_defaults(data)
                    self._skip_flow_training = self.flow is not None
            try:
                yield self
            finally:
                for a in attrs:
                    if a in saved:
                        setattr(self, a, saved[a])
                    elif hasattr(self, a):
                        delattr(self, a)

        return ctx()

    # ---- configuration / convenience (aspire.py:762-909) ---------------------------------------
    def config_dict(self, include_sampler_config: bool = False, **kwargs) -> dict:
        def fid(f):
            return f"{getattr(f, '__module__', None)}.{getattr(f, '__qualname__', getattr(f, '__name__', repr(f)))}"

        config = {
            "log_likelihood": fid(self.log_likelihood), "log_prior": fid(self.log_prior), "dims": self.dims,
            "parameters": self.parameters, "periodic_parameters": self.periodic_parameters,
            "prior_bounds": self.prior_bounds, "bounded_to_unbounded": self.bounded_to_unbounded,
            "bounded_transform": self.bounded_transform, "flow_matching": self.flow_matching, "device": self.device,
            "xp": self.xp.__name__ if self.xp else None, "flow_backend": self.flow_backend,
            "flow_kwargs": self.flow_kwargs, "eps": self.eps,
        }
        if include_sampler_config:
            if hasattr(self, "_last_sampler_type"):
                config["sampler_type"] = self._last_sampler_type
            if self.sampler is None:
                raise ValueError("Sampler has not been initialized.")
            config["sampler_config"] = self.sampler.config_dict(**kwargs)
        return config

    def save_config_to_json(self, filename: str) -> None:
        import json

        with open(filename, "w") as f:
            json.dump(self.config_dict(), f, indent=4, default=str)

    def sample_flow(self, n_samples: int = 1, xp=None) -> Samples:
        """aspire.py:891-909: draws from the flow (data transform included), no prior / likelihood evaluation."""
        if self.flow is None:
            self.init_flow()
        x, log_q = self.flow.sample_and_log_prob(n_samples)
        out_xp = xp if xp is not None else (self.xp if self.xp is not None else np)
        if is_torch_namespace(out_xp):
            return Samples(x=x, log_q=log_q, xp=out_xp, parameters=self.parameters, dtype=self.dtype)
        return Samples(x=to_numpy(x), log_q=to_numpy(log_q), xp=np, parameters=self.parameters, dtype=self.dtype)

