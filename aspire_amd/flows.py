"""Proposal flows behind the reference's `Flow` interface (src/aspire/flows/base.py:11-98).

`GaussianFlow`  — analytic diagonal-Gaussian proposal; sampling and log-density run in HIP kernels
                  (asmc_gaussian_draw / asmc_mixture_logpdf) and it exposes `device_mixture()` so the
                  pCN kernel can evaluate log_q in-register (fused path).
`CouplingFlow`  — RealNVP-style affine coupling flow in PyTorch-ROCm (the north-star's "PyTorch only
                  for the flow forward/inverse pass"): stands in for the reference's zuko MAF
                  (src/aspire/flows/torch/flows.py:113-444; zuko is absent from this image, so the
                  flow arithmetic is this repository's own — parity unpinned).
"""
from __future__ import annotations

import contextlib
import logging
import math

import numpy as np
import torch

from .history import FlowHistory

logger = logging.getLogger(__name__)


class Flow:
    """flows/base.py:11-98: the interface the samplers rely on."""

    xp = None

    def __init__(self, dims: int, device=None, data_transform=None):
        self.dims = dims
        self.device = device
        self.data_transform = data_transform  # None = identity (flows/base.py:36-40)

    # flows/base.py:54-61 — the flow lives in the data transform's space
    def _has_transform(self) -> bool:
        t = self.data_transform
        return t is not None and not getattr(t, "is_identity", False)

    def fit_data_transform(self, x):
        return self.data_transform.fit(x) if self._has_transform() else x

    def rescale(self, x):
        """x -> (x', log|det dx'/dx|)."""
        return self.data_transform.forward(x) if self._has_transform() else (x, None)

    def inverse_rescale(self, x_prime):
        """x' -> (x, log|det dx/dx'|)."""
        return self.data_transform.inverse(x_prime) if self._has_transform() else (x_prime, None)

    def log_prob(self, x):
        raise NotImplementedError

    def sample(self, n_samples):
        return self.sample_and_log_prob(n_samples)[0]

    def sample_and_log_prob(self, n_samples):
        raise NotImplementedError

    def fit(self, samples, **kwargs) -> FlowHistory:
        raise NotImplementedError

    def config_dict(self):
        return getattr(self, "_init_args", {})


class GaussianFlow(Flow):
    """q(x) = N(mu, diag sigma^2), device resident."""

    xp = torch

    def __init__(self, dims: int, mu=0.0, sigma=1.0, seed: int | None = None, device=None, dtype=torch.float64,
                 engine=None, data_transform=None):
        super().__init__(dims, device=device, data_transform=data_transform)
        self.mu = np.broadcast_to(np.asarray(mu, dtype=np.float64), (dims,)).copy()
        self.sigma = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (dims,)).copy()
        # seed=None: the sampler derives the Philox key of the draw stream from ITS generator at the start of every run
        # (in the reference flow sampling is stochastic per run, flows/torch/flows.py:327-346), so runs with different
        # rng seeds start from different populations; an explicit seed pins the stream
        self.seed_from_rng = seed is None
        self.seed = 1234 if seed is None else int(seed)
        self.dtype = dtype if isinstance(dtype, torch.dtype) else {"float32": torch.float32, "float64": torch.float64}[np.dtype(dtype).name]
        self.engine = engine
        self.gid0 = 0  # global index of this rank's first particle (sharded runs)
        self._draws = 0
        self._dev = None

    def _eng(self):
        if self.engine is None:
            from .samples import get_default_engine

            self.engine = get_default_engine()
        return self.engine

    def _device_params(self):
        e = self._eng()
        if self._dev is None or self._dev[0] is not e:
            logw = -np.sum(np.log(self.sigma)) - 0.5 * self.dims * math.log(2 * math.pi)
            self._dev = (e, e.asarray(self.mu), e.asarray(self.sigma),
                         e.make_mixture([logw], self.mu[None], (1.0 / self.sigma**2)[None]))
        return self._dev

    def device_mixture(self, engine=None):
        if self._has_transform():
            raise ValueError("the proposal is Gaussian in the data transform's space, not in x")
        if engine is not None:
            self.engine = engine
        return self._device_params()[3]

    def fit(self, samples, **kwargs) -> FlowHistory:
        """Moment-match the diagonal Gaussian to the training samples (closed form)."""
        x = np.asarray(samples.detach().cpu() if isinstance(samples, torch.Tensor) else samples, dtype=np.float64)
        if self._has_transform():
            if getattr(self.data_transform, "engine", None) is None and hasattr(self.data_transform, "engine"):
                self.data_transform.engine = self._eng()
            x = np.asarray(self.fit_data_transform(x), dtype=np.float64)
        self.mu = x.mean(axis=0)
        self.sigma = x.std(axis=0, ddof=1)
        self._dev = None
        return FlowHistory()

    def sample_and_log_prob(self, n_samples: int):
        e, mu, sigma, mix = self._device_params()
        x, _ = e.gaussian_draw(n_samples, self.dims, self.dtype, mu, sigma, self.seed, self.gid0, self._draws,
                               want_lq=False)
        self._draws += 1
        lq = e.mixture_logpdf(x, mix)
        if self._has_transform():  # flows/torch/flows.py:327-346: (x, log p(x') - log|det dx/dx'|)
            x, logj = self.inverse_rescale(x)
            lq = lq - e.asarray(logj)
            x = e.asarray(x, dtype=self.dtype)
        return x, lq

    def log_prob_from_preconditioned(self, T):
        """`(z, logj) -> log q(T^-1(z))` for the preconditioning transform `T` of the chain (`logj` = log|det dT^-1/dz| at z,
        which the chain has anyway), WITHOUT the round trip z -> x -> this flow's latent through the bounded stage's
        special functions; None when the shortcut does not apply.

        It applies when `T` and this flow's data transform share their bounded -> unbounded stage (same logit or probit
        coordinates, bounds and eps, nothing periodic - what `Aspire(prior_bounds=...)` builds for both).  Then
        x = P^-1(y) with y = z s_T + m_T, and the flow's latent is w = (t - m_f) / s_f with t = P(x) = clip(y): the clip is
        the forward transform's eps clamp of the unit interval.  log q(x) = log N(w; mu, sigma^2) - sum log s_f + log|dP/dx|
        (flows/torch/flows.py:368-387 behind transforms.py:270-316) is a Gaussian in t - one premapped pass over z,
        asmc_mixture_logpdf_premap - plus the bounded stage's log-Jacobian:
          logit  (utils.py:196-245): the inverse clamps the unit interval exactly like the forward, so
                 log|dP/dx| at x = -(log|dP^-1/dy| at y) = -(logj + aff_T), aff_T = T's affine log-Jacobian (0 without one);
          probit (transforms.py:540-571): the inverse does not clamp, so the term is written out,
                 sum_bounded (0.5 log 2 pi + 0.5 t^2 - log(up - lo)), and rides on the same pass.
        The reference's own round trip loses digits of y in the tails (u next to 0 or 1); this form does not."""
        from scipy.special import erfinv

        Tf = self.data_transform
        if not self._has_transform() or self.mu is None:
            return None
        need = ("_kind", "_periodic", "_lower", "_upper", "_mean", "_std", "eps", "affine_transform")
        if any(not hasattr(t, k) for t in (T, Tf) for k in need):
            return None
        kind = np.asarray(Tf._kind)
        b = kind != 0
        kinds = set(kind[b].tolist())
        if (not np.array_equal(kind, np.asarray(T._kind)) or len(kinds) > 1 or np.any(Tf._periodic) or np.any(T._periodic)
                or Tf.eps != T.eps or not np.array_equal(Tf._lower[b], T._lower[b])
                or not np.array_equal(Tf._upper[b], T._upper[b])):
            return None
        if (T.affine_transform and T._mean is None) or (Tf.affine_transform and Tf._mean is None):
            return None
        e = self._eng()
        if not hasattr(e, "mixture_logpdf_premap"):
            return None
        d = self.dims
        one, zero = np.ones(d), np.zeros(d)
        s_t, m_t = (np.asarray(T._std), np.asarray(T._mean)) if T.affine_transform else (one, zero)
        s_f, m_f = (np.asarray(Tf._std), np.asarray(Tf._mean)) if Tf.affine_transform else (one, zero)
        probit = kinds == {2}
        eps = float(Tf.eps)
        if probit:
            t_lo, t_hi = (float(np.sqrt(2.0) * erfinv(2.0 * u - 1.0)) for u in (eps, 1.0 - eps))
        else:  # logit of the clamped unit interval, as the forward transform forms it: log(u) - log1p(-u)
            t_lo, t_hi = (float(np.log(u) - np.log1p(-u)) for u in (eps, 1.0 - eps))
        premap = np.stack([s_t, m_t, np.where(b, t_lo, -np.inf), np.where(b, t_hi, np.inf),
                           np.where(b, 0.5, 0.0) if probit else zero])
        sig = np.asarray(self.sigma, dtype=np.float64) * s_f
        mean = np.asarray(self.mu, dtype=np.float64) * s_f + m_f
        const = -np.log(sig).sum() - 0.5 * d * np.log(2.0 * np.pi)
        if probit:
            const += b.sum() * 0.5 * np.log(2.0 * np.pi) - np.log(Tf._upper[b] - Tf._lower[b]).sum()
        elif T.affine_transform:
            const += np.log(s_t).sum()  # - aff_T
        mix = e.make_mixture([const], mean[None], (1.0 / sig**2)[None])
        pm = e.asarray(np.ascontiguousarray(premap, dtype=np.float64))
        minus = not (probit or not b.any())

        def logq(z, logj):
            out = e.mixture_logpdf_premap(z, pm, mix)
            return out - logj if minus else out

        logq.fused_args = (pm, mix, minus)  # for asmc_pcn_ysplit_propose_tr, which evaluates this inside the propose kernel
        return logq

    def log_prob(self, x):
        e, _, _, mix = self._device_params()
        xt = e.asarray(x, dtype=x.dtype if isinstance(x, torch.Tensor) and x.dtype in (torch.float32, torch.float64) else torch.float64)
        if self._has_transform():  # flows/torch/flows.py:368-387: log p(x') + log|det dx'/dx|
            xp_, logj = self.rescale(xt)
            return e.mixture_logpdf(e.asarray(xp_, dtype=xt.dtype), mix) + e.asarray(logj)
        return e.mixture_logpdf(xt, mix)


class _Coupling(torch.nn.Module):
    def __init__(self, dims, mask, hidden):
        super().__init__()
        self.register_buffer("mask", mask)
        n_in = int(mask.sum().item())
        n_out = dims - n_in
        layers, last = [], n_in
        for h in hidden:
            layers += [torch.nn.Linear(last, h), torch.nn.ReLU()]
            last = h
        layers.append(torch.nn.Linear(last, 2 * n_out))
        self.net = torch.nn.Sequential(*layers)
        torch.nn.init.zeros_(self.net[-1].weight)
        torch.nn.init.zeros_(self.net[-1].bias)
        self.n_out = n_out

    def _st(self, x):
        h = self.net(x[:, self.mask])
        s, t = h[:, : self.n_out], h[:, self.n_out:]
        return 2.0 * torch.tanh(s / 2.0), t  # bounded log-scale

    def forward(self, x):  # data -> latent
        s, t = self._st(x)
        z = x.clone()
        z[:, ~self.mask] = (x[:, ~self.mask] - t) * torch.exp(-s)
        return z, -s.sum(-1)

    def inverse(self, z):  # latent -> data
        s, t = self._st(z)
        x = z.clone()
        x[:, ~self.mask] = z[:, ~self.mask] * torch.exp(s) + t
        return x, s.sum(-1)


class CouplingFlow(Flow):
    """RealNVP affine-coupling flow (alternating half masks, MLP conditioners) in PyTorch.

    Interface mirrors ZukoFlow (flows/torch/flows.py:113-444): fit / sample_and_log_prob / log_prob /
    forward / inverse.  Dense layers run on hipBLASLt (MFMA) through torch; everything else in the
    SMC step stays in the HIP kernels.
    """

    xp = torch

    def __init__(self, dims: int, n_layers: int = 4, hidden_features=(64, 64), seed: int = 1234, device=None,
                 dtype=torch.float32, data_transform=None):
        super().__init__(dims, device=torch.device(device or "cpu"), data_transform=data_transform)
        self.dtype = dtype
        half = torch.arange(dims) < (dims + 1) // 2
        masks = [half if i % 2 == 0 else ~half for i in range(n_layers)]
        if dims == 1:
            raise ValueError("CouplingFlow needs dims >= 2")
        self._seed, self._rng_snapshot = int(seed), None
        with self._private_rng():  # initialisation draws from the flow's own stream: the process-wide torch RNG is left alone
            self.layers = torch.nn.ModuleList([_Coupling(dims, m, tuple(map(int, hidden_features))) for m in masks])
        self.layers.to(device=self.device, dtype=dtype)
        self.loc = torch.zeros(dims, device=self.device, dtype=dtype)
        self.scale = torch.ones(dims, device=self.device, dtype=dtype)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._packed, self._version = None, 0  # device pack cache; bumped whenever the parameters change
        self._init_args = dict(dims=int(dims), n_layers=int(n_layers), hidden_features=[int(h) for h in hidden_features],
                               seed=int(seed), dtype=str(dtype).replace("torch.", ""))

    @contextlib.contextmanager
    def _private_rng(self):
        """The flow's own torch RNG stream (weight initialisation, the shuffles of `fit`): seeded with the flow's seed on first
        use, carried from one use to the next, and run inside `torch.random.fork_rng`, so that building or training a flow
        never re-seeds or advances the process-wide generators (a user density drawing from torch's RNG saw repeated streams
        when every refit called `torch.manual_seed`).  The stream is the one `torch.manual_seed(seed)` used to start: trained
        parameters are unchanged."""
        cuda = [self.device] if self.device.type == "cuda" else []
        with torch.random.fork_rng(devices=cuda):
            if self._rng_snapshot is None:
                torch.manual_seed(self._seed)
            else:
                torch.set_rng_state(self._rng_snapshot[0])
                if cuda and self._rng_snapshot[1] is not None:
                    torch.cuda.set_rng_state(self._rng_snapshot[1], self.device)
            try:
                yield
            finally:
                self._rng_snapshot = (torch.get_rng_state(), torch.cuda.get_rng_state(self.device) if cuda else None)

    def export_layers(self):
        """(weights, biases): fp32 numpy arrays, three dense layers per coupling layer, torch Linear layout."""
        ws, bs = [], []
        for layer in self.layers:
            lin = [m for m in layer.net if isinstance(m, torch.nn.Linear)]
            if len(lin) != 3:
                raise ValueError("the HIP coupling kernel needs exactly two hidden layers")
            for m in lin:
                ws.append(m.weight.detach().to("cpu", torch.float32).numpy())
                bs.append(m.bias.detach().to("cpu", torch.float32).numpy())
        return ws, bs

    def device_coupling(self, engine):
        """Pack the flow for asmc_coupling_logprob (fp32 MFMA kernel).  Only for float32 flows (the kernel computes
        in the flow's dtype) of a supported shape; raises otherwise so callers fall back to the torch modules
        knowingly."""
        if self.dtype != torch.float32:
            raise ValueError("device_coupling needs a float32 flow")
        if self._has_transform():
            raise ValueError("the flow lives in the data transform's space; the kernel evaluates log q in x")
        key = (id(engine), self._version)
        if self._packed is None or self._packed[0] != key:
            ws, bs = self.export_layers()
            hidden = ws[0].shape[0]
            if ws[1].shape != (hidden, hidden) or self.dims % 2:
                raise ValueError("the HIP coupling kernel needs equal hidden widths and even dims")
            self._packed = (key, engine.make_coupling(self.dims, hidden, ws, bs, self.loc.detach().cpu().numpy(),
                                                     self.scale.detach().cpu().numpy()))
        return self._packed[1]

    def sync_shards(self, comm):
        """Sharded runs (one process per GPU): every rank must evaluate the SAME flow - training is not bit-reproducible
        across processes - and draw its shard from its own stream.  Rank 0's parameters go to everyone (one all-gather of
        the flattened parameters) and the latent generator is re-seeded per rank."""
        if not comm.sharded:
            return
        # A flow that has not been trained (or moved) since its last synchronisation over a group of this shape is the same on
        # every rank already: the ranks agree on that through one tiny exchange (every rank must take the same branch - what
        # follows is a collective) and skip the parameter broadcast, the re-seeding and the repacking of the device-side copy
        # that the version bump would cause (a sample() call per run: 2.4 ms and ~140 small launches, profiles/r04K_*).
        key = (int(comm.world), int(comm.rank))
        dirty = self.__dict__.get("_synced") != (key, self._version)
        if hasattr(comm, "all_gather_i64") and not bool(np.asarray(comm.all_gather_i64(np.array([int(dirty)], dtype=np.int64))).any()):
            return
        params = [p for p in self.layers.parameters()]
        flat = torch.cat([p.detach().reshape(-1).to(torch.float64) for p in params]
                         + [self.loc.detach().double().reshape(-1), self.scale.detach().double().reshape(-1)]).contiguous()
        root = comm.all_gather_tensor(flat)[: flat.numel()]
        off = 0
        with torch.no_grad():
            for p in params:
                p.copy_(root[off: off + p.numel()].reshape(p.shape).to(p.dtype))
                off += p.numel()
            self.loc = root[off: off + self.dims].to(self.dtype)
            self.scale = root[off + self.dims: off + 2 * self.dims].to(self.dtype)
        self._gen.manual_seed(int(self._gen.initial_seed()) + 1_000_003 * int(comm.rank))
        self._version += 1
        self._synced = (key, self._version)

    def to(self, device):
        self.device = torch.device(device)
        self.layers.to(self.device)
        self.loc, self.scale = self.loc.to(self.device), self.scale.to(self.device)
        g = torch.Generator(device=self.device)
        g.manual_seed(self._gen.initial_seed())
        self._gen = g
        return self

    def _to_latent(self, x):
        z = (x - self.loc) / self.scale
        ladj = -torch.log(self.scale).sum().expand(x.shape[0]).clone()
        for layer in self.layers:
            z, l = layer(z)
            ladj = ladj + l
        return z, ladj

    def _from_latent(self, z):
        x = z
        ladj = torch.zeros(z.shape[0], device=z.device, dtype=z.dtype)
        for layer in reversed(self.layers):
            x, l = layer.inverse(x)
            ladj = ladj + l
        return x * self.scale + self.loc, ladj + torch.log(self.scale).sum()

    def _base_logp(self, z):
        return -0.5 * (z * z).sum(-1) - 0.5 * self.dims * math.log(2 * math.pi)

    def fit(self, samples, n_epochs: int = 100, lr_annealing: bool = False, patience: int = 20, batch_size: int = 500,
            validation_fraction: float = 0.2, lr: float = 1e-3, clip_grad: float | None = None, **kwargs) -> FlowHistory:
        """Maximum-likelihood training, same knobs as ZukoFlow.fit (flows/torch/flows.py:170-325)."""
        xs = np.asarray(samples.detach().cpu() if isinstance(samples, torch.Tensor) else samples)
        if self._has_transform():
            xs = np.asarray(self.fit_data_transform(np.asarray(xs, dtype=np.float64)))
        x = torch.as_tensor(xs, dtype=self.dtype, device=self.device)
        self.loc, self.scale = x.mean(0), x.std(0).clamp_min(1e-6)
        n = x.shape[0]
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(0)).to(self.device)
        n_val = int(n * validation_fraction)
        xv, xt = x[perm[:n_val]], x[perm[n_val:]]
        opt = torch.optim.Adam(self.layers.parameters(), lr=lr)
        hist = FlowHistory()
        best, best_state, bad = float("inf"), None, 0
        with self._private_rng():
            best, best_state = self._train(opt, hist, xt, xv, n_val, n_epochs, batch_size, patience, clip_grad)
        if best_state is not None:
            self.layers.load_state_dict(best_state)
        self.layers.eval()
        self._version += 1
        return hist

    def _train(self, opt, hist, xt, xv, n_val, n_epochs, batch_size, patience, clip_grad):
        best, best_state, bad = float("inf"), None, 0
        for _ in range(n_epochs):
            self.layers.train()
            order = torch.randperm(xt.shape[0], device=self.device)
            tot = 0.0
            for i in range(0, xt.shape[0], batch_size):
                xb = xt[order[i:i + batch_size]]
                z, ladj = self._to_latent(xb)
                loss = -(self._base_logp(z) + ladj).mean()
                opt.zero_grad()
                loss.backward()
                if clip_grad is not None:
                    torch.nn.utils.clip_grad_norm_(self.layers.parameters(), clip_grad)
                opt.step()
                tot += float(loss.detach()) * xb.shape[0]
            hist.training_loss.append(tot / max(1, xt.shape[0]))
            if n_val:
                with torch.no_grad():
                    z, ladj = self._to_latent(xv)
                    vl = float(-(self._base_logp(z) + ladj).mean())
                hist.validation_loss.append(vl)
                if vl < best:
                    best, bad = vl, 0
                    best_state = {k: v.detach().clone() for k, v in self.layers.state_dict().items()}
                else:
                    bad += 1
                    if bad >= patience:
                        break
        return best, best_state

    @torch.no_grad()
    def attach_engine(self, engine, sample_dtype=None, gid0: int = 0):
        """Let `sample_and_log_prob` draw on `engine` (asmc_coupling_sample: latent draws from the counter-based generator,
        inverted coupling layers on the matrix cores, x emitted in `sample_dtype`); `gid0` = global index of this rank's
        first particle.  Without an engine, or for shapes the kernel does not take, the torch modules sample."""
        self.engine, self.sample_dtype, self.gid0 = engine, sample_dtype, int(gid0)

    def _sample_on_engine(self, n_samples: int):
        e = getattr(self, "engine", None)
        if (e is None or not hasattr(e, "coupling_sample") or self.dtype != torch.float32 or self._has_transform()
                or getattr(self, "_engine_sampling_off", False)):
            return None
        try:
            dev = self.device_coupling(e)
            self._hip_draws = getattr(self, "_hip_draws", 0) + 1
            return e.coupling_sample(n_samples, getattr(self, "sample_dtype", None) or self.dtype, dev,
                                     int(self._gen.initial_seed()), getattr(self, "gid0", 0), self._hip_draws)
        except Exception as exc:  # unsupported shape / fp32-MFMA mode: the torch modules from here on
            logger.info("flow samples with its torch modules: %s", exc)
            self._engine_sampling_off = True
            return None

    @torch.no_grad()
    def sample_and_log_prob(self, n_samples: int, xp=None):
        out = self._sample_on_engine(n_samples)
        if out is not None:
            return out
        z = torch.randn((n_samples, self.dims), device=self.device, dtype=self.dtype, generator=self._gen)
        x, ladj = self._from_latent(z)
        lq = self._base_logp(z) - ladj
        if self._has_transform():  # flows/torch/flows.py:327-346
            x, logj = self.inverse_rescale(x.double())
            x = torch.as_tensor(x, device=self.device)
            lq = lq.double() - torch.as_tensor(logj, device=self.device)
        return x, lq

    @torch.no_grad()
    def log_prob(self, x, xp=None):
        if self._has_transform():  # flows/torch/flows.py:368-387
            xp_, logj = self.rescale(torch.as_tensor(x, device=self.device).double())
            z, ladj = self._to_latent(torch.as_tensor(xp_, dtype=self.dtype, device=self.device))
            return (self._base_logp(z) + ladj).double() + torch.as_tensor(logj, device=self.device)
        x = torch.as_tensor(x, dtype=self.dtype, device=self.device)
        z, ladj = self._to_latent(x)
        return self._base_logp(z) + ladj

    @torch.no_grad()
    def forward(self, x, xp=None):
        return self._to_latent(torch.as_tensor(x, dtype=self.dtype, device=self.device))

    @torch.no_grad()
    def log_prob_f64(self, x):
        """log q(x) with the SAME parameters widened to fp64 and fp64 arithmetic throughout (on the flow's device): the
        yardstick the accuracy of the fp32 / split-fp16 kernels is quoted against (bench.py `flow_max_rel_vs_fp64`)."""
        import copy

        x = torch.as_tensor(x, device=self.device).to(torch.float64)
        layers = copy.deepcopy(self.layers).to(torch.float64)
        z = (x - self.loc.double()) / self.scale.double()
        ladj = -torch.log(self.scale.double()).sum().expand(x.shape[0]).clone()
        for layer in layers:
            z, l = layer(z)
            ladj = ladj + l
        return self._base_logp(z) + ladj

    # flows/torch/flows.py:63-110: <path>/config (one dataset per flattened key, utils.py:841-887) and <path>/weights (one
    # dataset per state-dict entry).  `h5_file` is an open h5py File / Group or anything with the same group protocol
    # (create_group, create_dataset, [] / items / in); this package does not import h5py itself.
    def save(self, h5_file, path="flow"):
        grp = h5_file.create_group(path)
        if self._has_transform():  # flows/torch/flows.py:77-78: the data transform travels in `<path>/data_transform`
            if not hasattr(self.data_transform, "save"):
                raise NotImplementedError("the flow's data transform cannot be saved (no `save`)")
            self.data_transform.save(grp, "data_transform")
        cfg = grp.create_group("config")
        for key, value in self._init_args.items():
            cfg.create_dataset(key, data=np.asarray(value) if not isinstance(value, str) else value)
        w = grp.create_group("weights")
        for name, tensor in self.layers.state_dict().items():
            w.create_dataset(name, data=tensor.detach().cpu().numpy())
        w.create_dataset("_loc", data=self.loc.detach().cpu().numpy())
        w.create_dataset("_scale", data=self.scale.detach().cpu().numpy())

    @classmethod
    def load(cls, h5_file, path="flow", device=None):
        grp = h5_file[path]

        def plain(v):
            v = v[()] if hasattr(v, "shape") or hasattr(v, "dtype") else v
            if isinstance(v, bytes):
                v = v.decode()
            return v.tolist() if isinstance(v, np.ndarray) else (v.item() if isinstance(v, np.generic) else v)

        cfg = {k: plain(v) for k, v in grp["config"].items()}
        cfg["dtype"] = getattr(torch, str(cfg["dtype"]))
        if "data_transform" in grp:  # flows/torch/flows.py:94-102
            from .transforms import CompositeTransform

            cfg["data_transform"] = CompositeTransform.load(grp, "data_transform", strict=False)
        obj = cls(device=device, **cfg)
        weights = {name: torch.as_tensor(np.asarray(d[()])) for name, d in grp["weights"].items()}
        obj.loc = weights.pop("_loc").to(device=obj.device, dtype=obj.dtype)
        obj.scale = weights.pop("_scale").to(device=obj.device, dtype=obj.dtype)
        obj.layers.load_state_dict(weights)
        obj.layers.eval()
        obj._version += 1
        return obj

    @torch.no_grad()
    def inverse(self, z, xp=None):
        return self._from_latent(torch.as_tensor(z, dtype=self.dtype, device=self.device))


_LN_1000 = 6.907755278982137  # -ln(slope) of zuko's MonotonicAffineTransform, slope = 1e-3


class _MaskedLinear(torch.nn.Linear):
    """Linear layer whose weight is multiplied by a fixed 0/1 mask (MADE, Germain et al. 2015)."""

    def __init__(self, n_in, n_out, mask):
        super().__init__(n_in, n_out)
        self.register_buffer("mask", mask)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight * self.mask, self.bias)


class _Autoregressive(torch.nn.Module):
    """One masked autoregressive affine transform: z_i = (x_i - t_i(x_<i)) exp(-s_i(x_<i)) in the layer's variable order
    (`order[k]` = position of variable k; reversed on every other layer).  data -> latent is one pass of the masked MLP,
    latent -> data one pass per variable."""

    def __init__(self, dims, hidden, reverse, affine="tanh"):
        super().__init__()
        self.affine = affine
        deg_in = torch.arange(1, dims + 1)
        if reverse:
            deg_in = deg_in.flip(0)
        self.register_buffer("order", deg_in)
        layers, deg_prev = [], deg_in
        for h in hidden:
            deg_h = (torch.arange(h) % max(dims - 1, 1)) + 1
            layers += [_MaskedLinear(len(deg_prev), h, (deg_h[:, None] >= deg_prev[None, :]).to(torch.get_default_dtype())),
                       torch.nn.ReLU()]
            deg_prev = deg_h
        deg_out = torch.cat([deg_in, deg_in])  # (log-scale, shift) of variable k see variables of strictly lower degree
        layers.append(_MaskedLinear(len(deg_prev), 2 * dims, (deg_out[:, None] > deg_prev[None, :]).to(torch.get_default_dtype())))
        self.net = torch.nn.Sequential(*layers)
        torch.nn.init.zeros_(self.net[-1].weight)
        torch.nn.init.zeros_(self.net[-1].bias)
        self.dims = dims

    def _st(self, x):
        h = self.net(x)
        if self.affine == "zuko":
            # zuko's MonotonicAffineTransform (slope 1e-3): z = x exp(ls) + shift with the soft-clipped log-scale
            # ls = raw / (1 + |raw| / ln 1000)  -  written in this module's convention z = (x - t) exp(-s): s = -ls, t = -shift exp(s)
            raw, shift = h[:, : self.dims], h[:, self.dims:]
            s = -raw / (1.0 + raw.abs() / _LN_1000)
            return s, -shift * torch.exp(s)
        return 2.0 * torch.tanh(h[:, : self.dims] / 2.0), h[:, self.dims:]  # bounded log-scale, as the coupling layers

    def forward(self, x):  # data -> latent
        s, t = self._st(x)
        return (x - t) * torch.exp(-s), -s.sum(-1)

    def inverse(self, z):  # latent -> data: variable of degree 1 first
        x = torch.zeros_like(z)
        for k in torch.argsort(self.order).tolist():
            s, t = self._st(x)
            x[:, k] = z[:, k] * torch.exp(s[:, k]) + t[:, k]
        s, _ = self._st(x)
        return x, s.sum(-1)


class MAFFlow(CouplingFlow):
    """Masked autoregressive flow (Papamakarios et al. 2017) - the flow class the reference asks zuko for by default
    (`ZukoFlow(flow_class="MAF")`, flows/torch/flows.py:140-164; zuko is absent from this image, so this is the repository's
    own statement of the architecture: `n_transforms` MADE transforms with alternating variable order, affine with a bounded
    log-scale).  Training, `log_prob`, `forward` / `inverse`, save / load are CouplingFlow's PyTorch code.  In the hot path the
    flow runs on the HIP kernels (`device_coupling` -> include/asmc.h ASMC_FLOW_MAF): the mutation's log q inside
    `asmc_pcn_mutate_flow` (one fused kernel per step at dims = 32) and the proposal draw in `asmc_coupling_sample` - no torch op
    inside the mutation loop."""

    def __init__(self, dims: int, n_transforms: int = 3, hidden_features=(64, 64), seed: int = 1234, device=None,
                 dtype=torch.float32, data_transform=None, affine: str = "tanh"):
        Flow.__init__(self, dims, device=torch.device(device or "cpu"), data_transform=data_transform)
        self.dtype = dtype
        self._seed, self._rng_snapshot = int(seed), None
        if affine not in ("tanh", "zuko"):
            raise ValueError(f"affine must be 'tanh' or 'zuko', got {affine!r}")
        self.affine = affine
        hidden = tuple(map(int, hidden_features))
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            with self._private_rng():
                self.layers = torch.nn.ModuleList([_Autoregressive(dims, hidden, reverse=bool(i % 2), affine=affine)
                                                   for i in range(n_transforms)])
        finally:
            torch.set_default_dtype(prev)
        self.layers.to(device=self.device, dtype=dtype)
        self.loc = torch.zeros(dims, device=self.device, dtype=dtype)
        self.scale = torch.ones(dims, device=self.device, dtype=dtype)
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)
        self._packed, self._version = None, 0
        self._init_args = dict(dims=int(dims), n_transforms=int(n_transforms), hidden_features=[int(h) for h in hidden],
                               seed=int(seed), dtype=str(dtype).replace("torch.", ""), affine=str(affine))

    @classmethod
    def from_zuko_state_dict(cls, state_dict, device=None, dtype=torch.float32):
        """A flow the REFERENCE trained - `zuko.flows.MAF(features, 0, transforms=T, hidden_features=(h1, h2))`, what
        `ZukoFlow(flow_class="MAF")` builds (`src/aspire/flows/torch/flows.py:156-168` of the reference) and `BaseTorchFlow.save`
        stores entry by entry (`flows.py:63-87`: `flow/weights/<state-dict key>`) - as an `MAFFlow` that runs on the HIP kernels.

        **UNVERIFIED**: zuko is not installed in the build image and no file of it is in the reference tree, so this follows zuko's
        documented module layout and arithmetic (v1.x) and could only be tested against a restatement of that documentation
        (`tests/test_host_logic.py::test_zuko_state_dict_adapter_...`), never against the package:
        * keys `transform.transforms.<i>.hyper.<2 k>.{weight, bias, mask}`, k = 0, 1, 2 (a `MaskedMLP`: `MaskedLinear`s at the
          even positions of a Sequential, ReLU between them), `F.linear(x, mask * weight, bias)`; the buffer
          `transform.transforms.<i>.order` and the base distribution's buffers are not needed (the variable order lives in the masks);
        * the last layer emits, per feature i, `(shift_i, scale_i)` in rows `2 i`, `2 i + 1`
          (`phi.unflatten(-1, (-1, 2))`); `MonotonicAffineTransform(shift, scale, slope=1e-3)` maps
          `z = x * exp(scale / (1 + |scale / log(slope)|)) + shift` with that exponent as the log-determinant.
        The rows are regrouped to this class's `[scale_0 .. scale_{d-1} | shift_0 .. shift_{d-1}]` and the transforms evaluate the
        soft-clipped form (`affine="zuko"`; device side `asmc_coupling.affine = ASMC_AFFINE_SOFTCLIP`).  The reference's flows carry
        no standardisation of their own (`loc = 0`, `scale = 1`); a data transform, if any, stays with the caller."""
        import re

        pat = re.compile(r"(?:^|\.)transforms\.(\d+)\.hyper\.(\d+)\.(weight|bias|mask)$")
        found: dict = {}
        for key, value in state_dict.items():
            m = pat.search(key)
            if m:
                found.setdefault(int(m.group(1)), {}).setdefault(int(m.group(2)), {})[m.group(3)] = torch.as_tensor(np.asarray(value) if not isinstance(value, torch.Tensor) else value)
        if not found:
            raise ValueError("no `transforms.<i>.hyper.<k>.{weight,bias,mask}` entries: not a zuko MAF state dict")
        n_tr = max(found) + 1
        lin0 = [found[0][k] for k in sorted(found[0])]
        if len(lin0) != 3:
            raise ValueError(f"expected two hidden layers (three masked linear layers per transform), found {len(lin0)}")
        dims = int(lin0[0]["weight"].shape[1])
        hidden = (int(lin0[0]["weight"].shape[0]), int(lin0[1]["weight"].shape[0]))
        if tuple(lin0[2]["weight"].shape) != (2 * dims, hidden[1]):
            raise ValueError("the last layer must emit (shift, scale) per feature; flows with a context are not supported")
        flow = cls(dims, n_transforms=n_tr, hidden_features=hidden, device=device, dtype=dtype, affine="zuko")
        regroup = torch.cat([torch.arange(1, 2 * dims, 2), torch.arange(0, 2 * dims, 2)])  # [scale rows | shift rows]
        with torch.no_grad():
            for i in range(n_tr):
                if i not in found or len(found[i]) != 3:
                    raise ValueError(f"transform {i} is incomplete")
                ours = [m for m in flow.layers[i].net if isinstance(m, _MaskedLinear)]
                for j, (k, m) in enumerate(zip(sorted(found[i]), ours)):
                    w, b, mask = found[i][k]["weight"], found[i][k]["bias"], found[i][k].get("mask")
                    if mask is None:
                        mask = torch.ones_like(w)
                    if j == 2:
                        w, b, mask = w[regroup], b[regroup], mask[regroup]
                    if tuple(w.shape) != tuple(m.weight.shape):
                        raise ValueError(f"transform {i}, layer {j}: shape {tuple(w.shape)}, expected {tuple(m.weight.shape)}")
                    m.weight.copy_(w.to(m.weight))
                    m.bias.copy_(b.to(m.bias))
                    m.mask.copy_(mask.to(m.mask))
                # the variable order of `inverse` (which coordinate is final after how many passes): degree of input k = 1 + the
                # number of inputs its shift / scale may depend on (read off the composed masks)
                dep = ((ours[2].mask[:dims] != 0).double() @ (ours[1].mask != 0).double() @ (ours[0].mask != 0).double()) > 0
                flow.layers[i].order.copy_((dep.sum(1) + 1).to(flow.layers[i].order))
        flow.layers.eval()
        flow._version += 1
        return flow

    def export_layers(self):
        """(weights, biases): the MASKED fp32 matrices (weight * mask) of the three dense layers of every transform, torch
        Linear layout - what `asmc_maf_pack` takes (the masks, and with them each transform's variable order, travel inside
        the weights as zeros)."""
        ws, bs = [], []
        for layer in self.layers:
            lin = [m for m in layer.net if isinstance(m, _MaskedLinear)]
            if len(lin) != 3:
                raise ValueError("the HIP autoregressive kernel needs exactly two hidden layers")
            for m in lin:
                ws.append((m.weight * m.mask).detach().to("cpu", torch.float32).numpy())
                bs.append(m.bias.detach().to("cpu", torch.float32).numpy())
        return ws, bs

    def device_coupling(self, engine):
        """Pack the flow for the HIP kernels (asmc_coupling_logprob / _sample / asmc_pcn_mutate_flow with kind = ASMC_FLOW_MAF).
        float32 flows of dims <= 128 with two equal hidden widths in {32, 64, 128}; raises otherwise so that callers fall back
        to the torch modules knowingly."""
        if self.dtype != torch.float32:
            raise ValueError("device_coupling needs a float32 flow")
        if self._has_transform():
            raise ValueError("the flow lives in the data transform's space; the kernel evaluates log q in x")
        key = (id(engine), self._version)
        if self._packed is None or self._packed[0] != key:
            ws, bs = self.export_layers()
            hidden = ws[0].shape[0]
            if ws[1].shape != (hidden, hidden):
                raise ValueError("the HIP autoregressive kernel needs equal hidden widths")
            self._packed = (key, engine.make_maf(self.dims, hidden, ws, bs, self.loc.detach().cpu().numpy(),
                                                 self.scale.detach().cpu().numpy(), affine=1 if self.affine == "zuko" else 0))
        return self._packed[1]
