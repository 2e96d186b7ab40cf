"""Preconditioning transforms on the device.

`CompositeTransform` mirrors the reference class of the same name (src/aspire/transforms.py:142-316):
periodic wrap -> bounded-to-unbounded (logit / probit of the unit interval) -> affine standardisation, with
`fit`, `forward`, `inverse`, `new_instance` and `config_dict`.  All per-particle arithmetic runs in the HIP
kernels `asmc_transform_forward` / `asmc_transform_inverse` (csrc/asmc_transform.hip); this class only keeps
the per-dimension tables and the two scalar Jacobian constants, which it computes with numpy exactly as the
reference does (`-log(upper - lower).sum()`, `-log|std|.sum()`).

Inputs may be device tensors (the sampler's case: results stay on the device) or host arrays (results come
back as numpy arrays); there is no host implementation — host arrays are uploaded, transformed on the GPU
and downloaded.
"""
from __future__ import annotations

import logging
from typing import Any

import numpy as np
import torch

from .comm import Comm

logger = logging.getLogger(__name__)

_KINDS = {"logit": 1, "probit": 2}


class CompositeTransform:
    def __init__(self, parameters: list, periodic_parameters: list | None = None,
                 prior_bounds: dict | None = None, bounded_to_unbounded: bool = True,
                 bounded_transform: str = "probit", affine_transform: bool = True, device=None, xp=None,
                 eps: float = 1e-6, dtype: Any = None, engine=None):
        # transforms.py:156-163 — same warnings / errors
        if prior_bounds is None:
            logger.warning("Missing prior bounds, some transforms may not be applied.")
        if periodic_parameters and not prior_bounds:
            raise ValueError("Must specify prior bounds to use periodic parameters.")
        if bounded_transform not in _KINDS:
            raise ValueError(f"Unknown bounded transform: {bounded_transform}")
        self.parameters = list(parameters)
        self.periodic_parameters = list(periodic_parameters or [])
        self.bounded_to_unbounded = bounded_to_unbounded
        self.bounded_transform = bounded_transform
        self.affine_transform = affine_transform
        self.eps = eps
        self.device = device
        self.xp = xp
        self.dtype = dtype
        self.engine = engine
        d = len(self.parameters)
        lower, upper = np.full(d, -np.inf), np.full(d, np.inf)
        if prior_bounds is None:
            self.prior_bounds = None
            self.bounded_parameters = None
        else:
            self.prior_bounds = {k: np.asarray(prior_bounds[k], dtype=np.float64) for k in self.parameters}
            for j, p in enumerate(self.parameters):
                lower[j], upper[j] = self.prior_bounds[p]
            if bounded_to_unbounded:  # transforms.py:183-189: finite bounds and not periodic
                self.bounded_parameters = [p for p in self.parameters
                                           if np.isfinite(self.prior_bounds[p]).all() and p not in self.periodic_parameters]
            else:
                self.bounded_parameters = None
        self._lower, self._upper = lower, upper
        self._periodic = np.array([p in self.periodic_parameters for p in self.parameters], dtype=np.int32)
        bounded = np.array([bool(self.bounded_parameters) and p in self.bounded_parameters for p in self.parameters])
        self._kind = np.where(bounded, _KINDS[bounded_transform], 0).astype(np.int32)
        if bounded.any():
            denom = upper[bounded] - lower[bounded]
            if np.any(denom == 0.0):  # transforms.py:513-518
                raise ValueError("Current floating precision (float64) is too small for specified parameter ranges")
            self._unit_logj = float(-np.log(denom).sum())
        else:
            self._unit_logj = 0.0
        wrap = self._periodic.astype(bool)
        if wrap.any() and not np.all(np.isfinite(lower[wrap]) & np.isfinite(upper[wrap])):
            raise ValueError("Periodic parameters need finite prior bounds.")
        self._mean = self._std = None
        self._affine_logj = 0.0
        self._dev = None  # (engine, DeviceTransform) cache

    # The reference composes stage objects (`_periodic_transform`, `_bounded_transform`, `_affine_transform`: transforms.py:192-236); here the
    # stages are the columns of ONE device table.  These read-only views answer what callers of the reference read off the stages (their
    # presence, `dtype`, the bounds, the fitted affine moments); a stage that is switched off is None, as there.
    def _stage_view(self, **kw):
        from types import SimpleNamespace

        return SimpleNamespace(dtype=self.dtype, xp=self.xp, **kw)

    @property
    def _periodic_transform(self):
        w = self._periodic.astype(bool)
        return self._stage_view(lower=self._lower[w], upper=self._upper[w]) if w.any() else None

    @property
    def _bounded_transform(self):
        b = self._kind != 0
        return self._stage_view(lower=self._lower[b], upper=self._upper[b], eps=self.eps, name=self.bounded_transform) if b.any() else None

    @property
    def _affine_transform(self):
        return self._stage_view(_mean=self._mean, _std=self._std) if self.affine_transform else None

    # ---- plumbing ---------------------------------------------------------------------------
    @property
    def is_identity(self) -> bool:
        return not (self._periodic.any() or self._kind.any() or self.affine_transform)

    def _eng(self):
        if self.engine is None:
            from .samples import get_default_engine

            self.engine = get_default_engine()
        return self.engine

    def _tables(self, with_affine: bool = True):
        e = self._eng()
        fitted = with_affine and self.affine_transform
        if fitted and self._mean is None:
            raise RuntimeError("the affine stage has not been fitted: call fit(x) first")
        key = (id(e), fitted, None if self._mean is None else self._mean.tobytes() + self._std.tobytes())
        if self._dev is None or self._dev[0] != key:
            # the kernel clamps the unit interval only where kind != 0 and wraps only where periodic: the infinite
            # bounds of untouched dimensions are never used, but keep the tables finite anyway
            lo = np.where(np.isfinite(self._lower), self._lower, 0.0)
            up = np.where(np.isfinite(self._upper), self._upper, 1.0)
            self._dev = (key, e.make_transform(self._kind, self._periodic, lo, up, self._mean if fitted else None,
                                               self._std if fitted else None, self.eps, self._unit_logj,
                                               self._affine_logj if fitted else 0.0))
        return e, self._dev[1]

    def _in(self, x):
        e = self._eng()
        host = not isinstance(x, torch.Tensor)
        xt = e.asarray(np.atleast_2d(np.asarray(x)) if host else (x if x.dim() == 2 else x.reshape(1, -1)),
                       dtype=x.dtype if (not host and x.dtype in (torch.float32, torch.float64)) else torch.float64)
        return xt, host

    def _out(self, y, lj, host):
        if not host:
            return y, lj
        e = self._eng()
        return e.to_numpy(y), (None if lj is None else e.to_numpy(lj))

    # ---- reference API (transforms.py:252-316) -----------------------------------------------
    def fit(self, x, comm: Comm | None = None):
        """Fit the affine stage to the (periodic + bounded transformed) data and return the transformed data.
        `comm`: sharded populations fit the GLOBAL mean / standard deviation."""
        xt, host = self._in(x)
        e, t0 = self._tables(with_affine=False)
        z0 = xt.clone() if self.is_identity else e.transform_forward(xt, t0, want_logj=False)[0]
        if self.affine_transform:
            comm = comm or Comm()
            n_glob = z0.shape[0]
            sums = np.asarray(e.colsum(z0), dtype=np.float64)
            if comm.sharded:  # shards may be ragged (owner-layout resampling): the row count travels with the sums
                parts = comm.all_gather_f64(np.concatenate([sums, [float(n_glob)]]))
                tot = parts[0].copy()
                for r in range(1, comm.world):
                    tot = tot + parts[r]
                sums, n_glob = tot[:-1], int(tot[-1])
            mean = sums / n_glob
            m2 = np.diag(np.asarray(e.centered_gram(z0, mean), dtype=np.float64)).copy()
            if comm.sharded:
                parts = comm.all_gather_f64(m2)
                m2 = parts[0].copy()
                for r in range(1, comm.world):
                    m2 = m2 + parts[r]
            self._mean, self._std = mean, np.sqrt(m2 / n_glob)  # x.mean(0), x.std(0) (population), transforms.py:622-626
            self._affine_logj = float(-np.log(np.abs(self._std)).sum())
            e, t = self._tables()
            z = e.transform_forward(xt, t, want_logj=False)[0]
        else:
            z = z0
        return self._out(z, None, host)[0]

    def forward(self, x):
        xt, host = self._in(x)
        e, t = self._tables()
        if self.is_identity:
            return self._out(xt.clone(), e.full(xt.shape[0], 0.0), host)
        return self._out(*e.transform_forward(xt, t), host)

    def inverse(self, z):
        zt, host = self._in(z)
        e, t = self._tables()
        if self.is_identity:
            return self._out(zt.clone(), e.full(zt.shape[0], 0.0), host)
        return self._out(*e.transform_inverse(zt, t), host)

    def new_instance(self, xp=None, dtype: Any = None):
        return self.__class__(parameters=self.parameters, periodic_parameters=self.periodic_parameters,
                              prior_bounds=self.prior_bounds, bounded_to_unbounded=self.bounded_to_unbounded,
                              bounded_transform=self.bounded_transform, affine_transform=self.affine_transform,
                              device=self.device, xp=xp or self.xp, eps=self.eps, dtype=dtype or self.dtype,
                              engine=self.engine)

    def config_dict(self):
        return {"xp": getattr(self.xp, "__name__", None), "dtype": str(self.dtype) if self.dtype else None,
                "parameters": self.parameters, "periodic_parameters": self.periodic_parameters,
                "prior_bounds": self.prior_bounds, "bounded_to_unbounded": self.bounded_to_unbounded,
                "bounded_transform": self.bounded_transform, "affine_transform": self.affine_transform,
                "eps": self.eps, "device": self.device}

    # transforms.py:63-122, 338-346, 639-646: a group with the class name as an attribute, `config` (dotted leaves) and the fitted
    # state of the affine stage (`affine_transform/mean`, `affine_transform/std`); `h5_file`: anything with h5py's group protocol
    def save(self, h5_file, path: str = "data_transform") -> None:
        from .io import recursively_save_to_h5_file

        grp = h5_file.create_group(path)
        grp.attrs["class"] = self.__class__.__name__
        cfg = self.config_dict()
        cfg["prior_bounds"] = None if cfg["prior_bounds"] is None else {k: [float(v[0]), float(v[1])] for k, v in cfg["prior_bounds"].items()}
        # the dtype as the reference encodes it (utils.py:544-565): {"__dtype__": True, "xp": <module name>, "dtype": <name>}
        dt_name = None if self.dtype is None else (str(self.dtype).split(".")[-1] if isinstance(self.dtype, torch.dtype) else np.dtype(self.dtype).name)
        cfg["dtype"] = None if dt_name is None else {"__dtype__": True, "xp": cfg["xp"], "dtype": dt_name}
        recursively_save_to_h5_file(grp, "config", cfg)
        if self.affine_transform and self._mean is not None:
            aff = grp.create_group("affine_transform")
            aff.create_dataset("mean", data=np.asarray(self._mean, dtype=np.float64))
            aff.create_dataset("std", data=np.asarray(self._std, dtype=np.float64))

    @classmethod
    def load(cls, h5_file, path: str = "data_transform", strict: bool = False, engine=None):
        from .io import load_from_h5_file

        grp = h5_file[path]
        name = grp.attrs["class"]
        name = name.decode() if isinstance(name, bytes) else str(name)
        known = {c.__name__: c for c in (CompositeTransform, FlowTransform)}
        if name != cls.__name__:
            if strict:
                raise ValueError(f"Expected class {cls.__name__}, got {name}.")
            if name not in known:
                raise ValueError(f"Unknown transform class {name}")
            cls = known[name]
        cfg = load_from_h5_file(grp, "config")
        xp_name = cfg.pop("xp", None)
        enc = cfg.pop("dtype", None)
        dt_name = enc.get("dtype") if isinstance(enc, dict) else (enc if isinstance(enc, str) else None)
        if isinstance(cfg.get("prior_bounds"), dict):
            cfg["prior_bounds"] = {k: (float(np.asarray(v)[0]), float(np.asarray(v)[1])) for k, v in cfg["prior_bounds"].items()}
        for key in ("parameters", "periodic_parameters"):
            if key in cfg and cfg[key] is not None:
                cfg[key] = [str(v) for v in np.asarray(cfg[key]).tolist()] if not isinstance(cfg[key], list) else cfg[key]
        if cls is FlowTransform:
            cfg.pop("periodic_parameters", None)
        for key in ("bounded_to_unbounded", "affine_transform"):
            if key in cfg:
                cfg[key] = bool(cfg[key])
        if "eps" in cfg:
            cfg["eps"] = float(cfg["eps"])
        xp = None
        if xp_name is not None:  # (the array namespace the file names, or plain numpy / torch where that wrapper package is absent)
            from ._xp import resolve_xp

            xp = resolve_xp(str(xp_name))
        dtype = None
        if dt_name is not None:
            from ._xp import resolve_dtype

            dtype = resolve_dtype(str(dt_name).split(".")[-1], xp if xp is not None else np)
        obj = cls(xp=xp, engine=engine, dtype=dtype, **cfg)
        if obj.affine_transform and "affine_transform" in grp:
            obj._mean = np.asarray(grp["affine_transform"]["mean"][()], dtype=np.float64)
            obj._std = np.asarray(grp["affine_transform"]["std"][()], dtype=np.float64)
            obj._affine_logj = float(-np.log(np.abs(obj._std)).sum())
        return obj


class FlowTransform(CompositeTransform):
    """The data transform the reference puts in front of its flows (transforms.py:345-395): a CompositeTransform
    without periodic parameters, affine stage on by default."""

    def __init__(self, parameters: list, prior_bounds: dict | None = None, bounded_to_unbounded: bool = True,
                 bounded_transform: str = "probit", affine_transform: bool = True, device=None, xp=None,
                 eps: float = 1e-6, dtype: Any = None, engine=None):
        super().__init__(parameters=parameters, periodic_parameters=[], prior_bounds=prior_bounds,
                         bounded_to_unbounded=bounded_to_unbounded, bounded_transform=bounded_transform,
                         affine_transform=affine_transform, device=device, xp=xp, eps=eps, dtype=dtype, engine=engine)

    def new_instance(self, xp=None, dtype: Any = None):
        return self.__class__(parameters=self.parameters, prior_bounds=self.prior_bounds,
                              bounded_to_unbounded=self.bounded_to_unbounded, bounded_transform=self.bounded_transform,
                              affine_transform=self.affine_transform, device=self.device, xp=xp or self.xp, eps=self.eps,
                              dtype=dtype or self.dtype, engine=self.engine)

    def config_dict(self):
        cfg = super().config_dict()
        cfg.pop("periodic_parameters", None)
        return cfg


class FlowPreconditioningTransform:
    """`preconditioning="flow"` (reference transforms.py:649-748): at every temperature a normalising flow is trained on the
    current particles and the Markov chain runs in its latent space, z = flow(T(x)) with T the composite data transform
    (periodic / bounded / affine stages) the reference puts in front of the flow.  The flow is one of this package's
    (`flow_backend` "coupling" or "maf"; zuko is absent); its passes and its training run in PyTorch, the composite stages on
    the engine.  `HipSMC._mutate_preconditioned` drives it through the split propose / accept kernels like any user transform."""

    def __init__(self, parameters: list, flow_backend: str = "coupling", prior_bounds: dict | None = None,
                 bounded_to_unbounded: bool = True, bounded_transform: str = "probit", affine_transform: bool = True,
                 periodic_parameters: list | None = None, device=None, xp=None, eps: float = 1e-6, dtype: Any = None,
                 flow_matching: bool = False, flow_kwargs: dict | None = None, fit_kwargs: dict | None = None, engine=None):
        if flow_matching:
            raise NotImplementedError("flow matching is out of scope (SURVEY.md §2)")
        self.parameters, self.periodic_parameters = parameters, periodic_parameters or []
        self.prior_bounds, self.bounded_to_unbounded, self.bounded_transform = prior_bounds, bounded_to_unbounded, bounded_transform
        self.affine_transform, self.eps, self.device, self.xp, self.dtype = affine_transform, eps, device, xp, dtype
        fc, fb = str((flow_kwargs or {}).get("flow_class", "")).upper(), str(flow_backend).lower()
        self.flow_backend = "maf" if (fc == "MAF" or fb == "maf") else "coupling" if fb in ("coupling", "zuko") else fb
        if self.flow_backend not in ("maf", "coupling"):
            raise ValueError(f"flow preconditioning needs a trainable flow (backend 'coupling' or 'maf'), not {flow_backend!r}")
        self.flow_matching = flow_matching
        self.flow_kwargs = {k: v for k, v in dict(flow_kwargs or {}).items() if k != "flow_class"}
        self.fit_kwargs = dict(fit_kwargs or {})
        self.engine = engine
        self._data_transform = CompositeTransform(parameters=parameters, periodic_parameters=periodic_parameters,
                                                  prior_bounds=prior_bounds, bounded_to_unbounded=bounded_to_unbounded,
                                                  bounded_transform=bounded_transform, affine_transform=affine_transform,
                                                  device=device, xp=xp, eps=eps, dtype=dtype, engine=engine)
        self.flow = None

    is_identity = False

    def _flow_device(self, like):
        import torch

        return like.device if isinstance(like, torch.Tensor) else torch.device(self.device or "cpu")

    def fit(self, x, comm: Comm | None = None):
        import torch

        from .flows import CouplingFlow, MAFFlow

        if self._data_transform.engine is None:
            self._data_transform.engine = self.engine
        u = self._data_transform.fit(x, comm=comm)
        ut = torch.as_tensor(u) if not isinstance(u, torch.Tensor) else u
        kw = dict(self.flow_kwargs)
        if "transforms" in kw:
            kw["n_transforms"] = int(kw.pop("transforms"))
        cls = MAFFlow if self.flow_backend == "maf" else CouplingFlow
        self.flow = cls(dims=len(self.parameters), device=self._flow_device(ut), dtype=kw.pop("flow_dtype", torch.float32), **kw)
        if comm is not None and comm.sharded:
            # the latent space is fitted to the WHOLE population, once: every rank contributes an equal strided share of at most
            # `fit_subsample` rows (all-gather, as the Student-t reference fit does), rank 0 trains on them, and its parameters
            # go to everyone - training is not bit-reproducible across processes, and every chain must run in the SAME space
            # `fit_subsample` (default 16384 pooled rows; None = every row of every rank) is a choice of THIS path: a single-rank
            # run trains on all its rows, as the reference does (transforms.py:700-716), so the two fit slightly different
            # problems unless fit_subsample=None.
            m = self.fit_kwargs.get("fit_subsample", 16384)
            k = ut.shape[0] if m is None else max(1, min(int(m) // comm.world, ut.shape[0]))
            rows = torch.as_tensor((np.arange(k, dtype=np.int64) * ut.shape[0]) // k, device=ut.device)
            pooled = comm.all_gather_ragged(ut[rows].contiguous(), [int(c) for c in comm.all_gather_i64(np.array([k]))[:, 0]])
            # the training can take longer than a collective may wait (watchdog timeouts): the other ranks block on a store key,
            # not inside sync_shards' first collective
            if comm.rank == 0:
                try:
                    self.flow.fit(pooled, **{a: b for a, b in self.fit_kwargs.items() if a != "fit_subsample"})
                except BaseException as exc:  # the waiting ranks raise too instead of sitting out the day-long timeout
                    comm.signal("flow_precond_fit", error=exc)
                    raise
                comm.signal("flow_precond_fit")
            else:
                comm.await_signal("flow_precond_fit")
            self.flow.sync_shards(comm)
        else:
            self.flow.fit(ut, **{a: b for a, b in self.fit_kwargs.items() if a != "fit_subsample"})
        return self._cast(self.flow.forward(ut)[0], u)

    def _cast(self, z, like):
        import torch

        if isinstance(like, torch.Tensor):
            return z.to(like.dtype)
        return z.double().cpu().numpy()

    def forward(self, x):
        import torch

        u, lj1 = self._data_transform.forward(x)
        ut = torch.as_tensor(u)
        z, lj2 = self.flow.forward(ut)
        if isinstance(u, torch.Tensor):
            return z.to(u.dtype), torch.as_tensor(lj1, device=z.device).double() + lj2.double()
        return z.double().cpu().numpy(), np.asarray(lj1) + lj2.double().cpu().numpy()

    def inverse(self, z):
        import torch

        zt = torch.as_tensor(z)
        u, lj2 = self.flow.inverse(zt)
        if isinstance(z, torch.Tensor):
            x, lj1 = self._data_transform.inverse(u.to(z.dtype))
            return x, torch.as_tensor(lj1, device=u.device).double() + lj2.double()
        x, lj1 = self._data_transform.inverse(u.double().cpu().numpy())
        return x, np.asarray(lj1) + lj2.double().cpu().numpy()

    def new_instance(self, xp=None, dtype: Any = None):
        return self.__class__(parameters=self.parameters, periodic_parameters=self.periodic_parameters,
                              prior_bounds=self.prior_bounds, bounded_to_unbounded=self.bounded_to_unbounded,
                              bounded_transform=self.bounded_transform, affine_transform=self.affine_transform,
                              device=self.device, xp=xp or self.xp, eps=self.eps, dtype=dtype or self.dtype,
                              flow_backend=self.flow_backend, flow_matching=self.flow_matching, flow_kwargs=self.flow_kwargs,
                              fit_kwargs=self.fit_kwargs, engine=self.engine)

    def save(self, h5_file, path="data_transform"):
        raise NotImplementedError("FlowPreconditioningTransform does not support save method yet.")  # transforms.py:745-748

    def config_dict(self):
        return {**self._data_transform.config_dict(), "flow_backend": self.flow_backend, "flow_kwargs": self.flow_kwargs,
                "fit_kwargs": self.fit_kwargs}
