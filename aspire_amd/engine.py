"""HipEngine — thin, 1:1 Python face of the C ABI (include/asmc.h) over torch device tensors.

torch is used only for device memory, streams and (in `comm.py`) torch.distributed; every
arithmetic step of the hot path runs in the hand-written HIP kernels of libasmc_hip.so.
There is no CPU fallback: constructing a HipEngine without a HIP device raises.

The same method set is implemented by the oracle-backed test double in `tests/oracle_engine.py`
so that the host logic (`smc_math.py`, the sampler loop) can be exercised without a GPU.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import (ASMC_CDF_EXACT, ASMC_CDF_FAST, ASMC_F32, ASMC_F64, AsmcCoupling, AsmcMixture, AsmcPcnParams,
                   AsmcTransform, check)

CDF_MODES = {"exact": ASMC_CDF_EXACT, "fast": ASMC_CDF_FAST}


def _dptr(t: torch.Tensor | None):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f64p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@dataclass
class DeviceMixture:
    """Diagonal Gaussian mixture log-density with parameters resident in HBM.

    log sum_c exp(logw[c] - 0.5 * sum_j (x_j - mu[c,j])^2 * prec[c,j]); logw includes constants.
    """

    logw: torch.Tensor  # [C]
    mu: torch.Tensor  # [C, d]
    prec: torch.Tensor  # [C, d]

    @property
    def n_components(self) -> int:
        return int(self.mu.shape[0])

    @property
    def dims(self) -> int:
        return int(self.mu.shape[1])

    def c_struct(self) -> AsmcMixture:
        return AsmcMixture(self.n_components, 0, self.logw.data_ptr(), self.mu.data_ptr(), self.prec.data_ptr())


@dataclass
class DeviceCoupling:
    """Affine coupling flow with its parameters packed in MFMA operand order, resident in HBM."""

    dims: int
    n_layers: int
    hidden: int
    packed: torch.Tensor  # fp32 [asmc_coupling_pack_floats / asmc_maf_pack_floats]
    loc: torch.Tensor  # fp32 [dims]
    scale: torch.Tensor  # fp32 [dims]
    log_scale_sum: float
    kind: int = 0  # ASMC_FLOW_COUPLING (0) / ASMC_FLOW_MAF (1)
    affine: int = 0  # ASMC_AFFINE_TANH (0) / ASMC_AFFINE_SOFTCLIP (1: zuko's monotonic affine form, autoregressive flows only)

    def c_struct(self) -> AsmcCoupling:
        return AsmcCoupling(self.dims, self.n_layers, self.hidden, self.kind, self.packed.data_ptr(), self.loc.data_ptr(),
                            self.scale.data_ptr(), self.log_scale_sum, self.affine, 0)


@dataclass
class DeviceTransform:
    """Per-dimension tables of a CompositeTransform in HBM (include/asmc.h `asmc_transform`)."""

    d: int
    kind: torch.Tensor  # int32 [d]: 0 none, 1 logit, 2 probit
    periodic: torch.Tensor  # int32 [d]
    lower: torch.Tensor  # fp64 [d]
    upper: torch.Tensor
    mean: torch.Tensor | None
    std: torch.Tensor | None
    eps: float
    unit_logj: float
    affine_logj: float
    hints: int = 0  # ASMC_TR_NO_* bits (include/asmc.h)

    def c_struct(self) -> AsmcTransform:
        return AsmcTransform(self.d, self.hints, self.kind.data_ptr(), self.periodic.data_ptr(), self.lower.data_ptr(),
                             self.upper.data_ptr(), None if self.mean is None else self.mean.data_ptr(),
                             None if self.std is None else self.std.data_ptr(), self.eps, self.unit_logj, self.affine_logj)


def pack_coupling(lib, dims: int, hidden: int, weights, biases) -> np.ndarray:
    """Host-side packing (asmc_coupling_pack): torch.nn.Linear-layout fp32 arrays, three per coupling layer."""
    n_layers = len(weights) // 3
    assert len(weights) == len(biases) == 3 * n_layers and n_layers >= 1
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    dh = dims // 2
    for c in range(n_layers):
        assert ws[3 * c].shape == (hidden, dh) and ws[3 * c + 1].shape == (hidden, hidden)
        assert ws[3 * c + 2].shape == (dims, hidden)
        assert bs[3 * c].shape == (hidden,) and bs[3 * c + 1].shape == (hidden,) and bs[3 * c + 2].shape == (dims,)
    nfl = lib.asmc_coupling_pack_floats(dims, n_layers, hidden)
    if nfl < 0:
        raise _lib.AsmcError(f"coupling flow shape not supported by the HIP kernel: {lib.asmc_last_error().decode()}")
    out = np.empty(nfl, dtype=np.float32)
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    check(lib.asmc_coupling_pack(dims, n_layers, hidden, wp, bp, out.ctypes.data_as(ctypes.c_void_p)), "asmc_coupling_pack")
    return out


def pack_maf(lib, dims: int, hidden: int, weights, biases) -> np.ndarray:
    """Host-side packing of a masked autoregressive flow (asmc_maf_pack): the MASKED dense layers, three per transform, in
    torch.nn.Linear layout ([hidden, dims], [hidden, hidden], [2 dims, hidden])."""
    n_tr = len(weights) // 3
    assert len(weights) == len(biases) == 3 * n_tr and n_tr >= 1
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    for c in range(n_tr):
        assert ws[3 * c].shape == (hidden, dims) and ws[3 * c + 1].shape == (hidden, hidden)
        assert ws[3 * c + 2].shape == (2 * dims, hidden)
        assert bs[3 * c].shape == (hidden,) and bs[3 * c + 1].shape == (hidden,) and bs[3 * c + 2].shape == (2 * dims,)
    nfl = lib.asmc_maf_pack_floats(dims, n_tr, hidden)
    if nfl < 0:
        raise _lib.AsmcError(f"autoregressive flow shape not supported by the HIP kernel: {lib.asmc_last_error().decode()}")
    out = np.empty(nfl, dtype=np.float32)
    wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
    check(lib.asmc_maf_pack(dims, n_tr, hidden, wp, bp, out.ctypes.data_as(ctypes.c_void_p)), "asmc_maf_pack")
    return out


class HipEngine:
    """One engine per (process, device).  Not thread-safe (the ctx is bound to one host thread)."""

    name = "hip"

    def __init__(self, device: int | str | torch.device = 0, n_max: int = 1 << 20, d_max: int = 32):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.AsmcError("HipEngine needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise _lib.AsmcError(f"HipEngine needs a cuda (HIP) device, got {dev}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.n_max, self.d_max = int(n_max), int(d_max)
        self._flow_work = None  # scratch of pcn_mutate_flow, grown on demand
        self._ctx = ctypes.c_void_p()
        check(self.lib.asmc_ctx_create(ctypes.byref(self._ctx), self.device.index, self.n_max, self.d_max), "asmc_ctx_create")

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.asmc_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ensure_capacity(self, n: int, d: int):
        if n > self.n_max or d > self.d_max:
            torch.cuda.synchronize(self.device)
            self.close()
            self.n_max, self.d_max = max(n, self.n_max), max(d, self.d_max)
            self._ctx = ctypes.c_void_p()
            check(self.lib.asmc_ctx_create(ctypes.byref(self._ctx), self.device.index, self.n_max, self.d_max), "asmc_ctx_create")

    # ---- per-kernel HIP-event timing --------------------------------------------------------
    def profile(self, on: bool):
        check(self.lib.asmc_profile_enable(self._ctx, int(on)), "asmc_profile_enable")

    def profile_report(self) -> dict:
        """{kernel: (launches, avg_ms)} since profiling was enabled (synchronises)."""
        buf = ctypes.create_string_buffer(1 << 16)
        check(self.lib.asmc_profile_report(self._ctx, buf, len(buf)), "asmc_profile_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.rsplit(" ", 2)
            out[name] = (int(cnt), float(ms))
        return out

    # ---- plumbing --------------------------------------------------------------------------
    @property
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def asarray(self, a, dtype=torch.float64) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device)

    def empty(self, shape, dtype=torch.float64) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def full(self, n: int, value: float) -> torch.Tensor:
        return torch.full((n,), value, dtype=torch.float64, device=self.device)

    def to_numpy(self, t: torch.Tensor) -> np.ndarray:
        return t.detach().cpu().numpy()

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()

    @staticmethod
    def _xdt(x: torch.Tensor) -> int:
        if x.dtype == torch.float64:
            return ASMC_F64
        if x.dtype == torch.float32:
            return ASMC_F32
        raise TypeError(f"x must be float64 or float32, got {x.dtype}")

    @staticmethod
    def _chk3(ll, lp, lq):
        for t in (ll, lp, lq):
            assert t.dtype == torch.float64 and t.is_contiguous() and t.dim() == 1

    # ---- weighting -------------------------------------------------------------------------
    def weights_max(self, ll, lp, lq, beta0: float, betas) -> tuple[np.ndarray, int]:
        self._chk3(ll, lp, lq)
        betas = np.ascontiguousarray(betas, dtype=np.float64)
        m = np.empty(betas.size)
        nnan = ctypes.c_int64(0)
        check(self.lib.asmc_weights_max(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, _f64p(betas),
                                        betas.size, _f64p(m), ctypes.byref(nnan), self._stream), "asmc_weights_max")
        return m, int(nnan.value)

    def weights_sums(self, ll, lp, lq, beta0: float, betas, m, shift=None) -> np.ndarray:
        self._chk3(ll, lp, lq)
        betas = np.ascontiguousarray(betas, dtype=np.float64)
        m = np.ascontiguousarray(m, dtype=np.float64)
        sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float64)
        out = np.empty(2 * betas.size)
        check(self.lib.asmc_weights_sums(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, _f64p(betas),
                                         _f64p(m), None if sh is None else _f64p(sh), betas.size, _f64p(out),
                                         self._stream), "asmc_weights_sums")
        return out.reshape(-1, 2)

    def weights_stats(self, ll, lp, lq, beta0: float, betas) -> np.ndarray:
        """[K,4] = (m, S1, S2, n_nan) per candidate beta, one pass over the batch."""
        self._chk3(ll, lp, lq)
        betas = np.ascontiguousarray(betas, dtype=np.float64)
        out = np.empty(4 * betas.size)
        check(self.lib.asmc_weights_stats(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0,
                                          _f64p(betas), betas.size, _f64p(out), self._stream), "asmc_weights_stats")
        return out.reshape(-1, 4)

    def find_beta(self, ll, lp, lq, beta0: float, target_eff: float, tol: float):
        """Device-side adaptive-beta search (single rank):
        (beta_star, eff_at_one, converged, rounds, n_nan, (m, S1, S2) at beta_star or None, (m, S1, S2) at 1)."""
        self._chk3(ll, lp, lq)
        out = np.zeros(13)
        check(self.lib.asmc_find_beta(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, target_eff, tol,
                                      _f64p(out), self._stream), "asmc_find_beta")
        trip = (float(out[6]), float(out[7]), float(out[8])) if out[9] != 0.0 else None
        return float(out[0]), float(out[4]), bool(out[2]), int(out[3]), int(out[5]), trip, tuple(map(float, out[10:13]))

    def importance_step(self, ll, lp, lq, beta0: float, target_eff: float, tol: float, state4: np.ndarray, n_out: int,
                        idx_out: torch.Tensor | None = None):
        """Enqueue one iteration's adaptive-beta search, evidence moments and resampling indices for the next `n_out`
        PCG64 doubles behind `state4` (include/asmc.h asmc_importance_step; no host synchronisation).  Returns the device
        index tensor; `importance_result()` then reads the scalars."""
        self._chk3(ll, lp, lq)
        n = ll.numel()
        st = np.ascontiguousarray(state4, dtype=np.uint64)
        bufs = self.__dict__.setdefault("_is_bufs", {})
        if bufs.get("n") != n:
            bufs.update(n=n, w=self.empty(n), cdf=self.empty(n))
        idx = idx_out if idx_out is not None else torch.empty(n_out, dtype=torch.int64, device=self.device)
        assert idx.dtype == torch.int64 and idx.numel() == n_out and idx.is_contiguous()
        check(self.lib.asmc_importance_step(self._ctx, n, _dptr(ll), _dptr(lp), _dptr(lq), beta0, target_eff, tol,
                                            st.ctypes.data_as(ctypes.c_void_p), n_out, _dptr(bufs["w"]), _dptr(bufs["cdf"]),
                                            _dptr(idx), self._stream), "asmc_importance_step")
        return idx

    def importance_result_enqueue(self):
        """Start `importance_result`'s read-back now (event behind it): work enqueued after this call does not delay it."""
        check(self.lib.asmc_importance_result_enqueue(self._ctx, self._stream), "asmc_importance_result_enqueue")

    def importance_result(self):
        """(beta_star, eff_at_one, converged, rounds, n_nan, (m, S1, S2) at beta_star or None, (m, S1, S2) at 1,
        m2, S1', found) of the last `importance_step` (synchronises)."""
        out = np.zeros(16)
        check(self.lib.asmc_importance_result(self._ctx, _f64p(out), self._stream), "asmc_importance_result")
        trip = (float(out[6]), float(out[7]), float(out[8])) if out[9] != 0.0 else None
        if not out[2] and not out[15] and self.lib.asmc_importance_available(self._ctx) == 0:
            self.importance_step_disabled = True  # the persistent kernel was not fully resident once: step-by-step from now on
        return (float(out[0]), float(out[4]), bool(out[2]), int(out[3]), int(out[5]), trip, tuple(map(float, out[10:13])),
                float(out[13]), float(out[14]), bool(out[15]))

    # sharded search (smc_math.find_beta_sharded drives the rounds; the all-gather between the halves is the caller's)
    def find_beta_shard_reduce(self, ll, lp, lq, beta0: float, rnd: int, rec: torch.Tensor):
        self._chk3(ll, lp, lq)
        assert rec.dtype == torch.float64 and rec.numel() >= _lib.ASMC_BIS_REC and rec.is_contiguous()
        check(self.lib.asmc_find_beta_shard_reduce(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, rnd,
                                                   _dptr(rec), self._stream), "asmc_find_beta_shard_reduce")

    def find_beta_shard_decide(self, recs: torch.Tensor, world: int, n_global: int, beta0: float, target_eff: float,
                               tol: float, rnd: int):
        assert recs.dtype == torch.float64 and recs.numel() == world * _lib.ASMC_BIS_REC and recs.is_contiguous()
        check(self.lib.asmc_find_beta_shard_decide(self._ctx, _dptr(recs), world, n_global, beta0, target_eff, tol, rnd,
                                                   self._stream), "asmc_find_beta_shard_decide")

    def find_beta_shard_rounds(self, comm, ll, lp, lq, beta0: float, target_eff: float, tol: float, n_global: int,
                               rec: torch.Tensor, recs: torch.Tensor, first: int, last: int):
        """Rounds first..last-1 (reduce -> all-gather -> decide) with the argument marshalling done once: the sharded
        search is bound by the host's enqueue rate, not by the GPU."""
        self._chk3(ll, lp, lq)
        lib, ctx, st = self.lib, self._ctx, self._stream
        a, b, c, r, rs = _dptr(ll), _dptr(lp), _dptr(lq), _dptr(rec), _dptr(recs)
        n, world, gather = ll.numel(), comm.world, comm.all_gather_into
        if last > first and self.use_rccl(comm):  # the library gathers the records itself: one call for all the rounds
            check(lib.asmc_find_beta_shard_rounds(ctx, n, a, b, c, beta0, target_eff, tol, world, n_global, r, rs, first, last, st),
                  "asmc_find_beta_shard_rounds")
            return
        for rnd in range(first, last):
            check(lib.asmc_find_beta_shard_reduce(ctx, n, a, b, c, beta0, rnd, r, st), "asmc_find_beta_shard_reduce")
            gather(recs, rec)
            check(lib.asmc_find_beta_shard_decide(ctx, rs, world, n_global, beta0, target_eff, tol, rnd, st),
                  "asmc_find_beta_shard_decide")

    def find_beta_shard_result(self):
        out = np.zeros(13)
        check(self.lib.asmc_find_beta_shard_result(self._ctx, _f64p(out), self._stream), "asmc_find_beta_shard_result")
        trip = (float(out[6]), float(out[7]), float(out[8])) if out[9] != 0.0 else None
        return float(out[0]), float(out[4]), bool(out[2]), int(out[3]), int(out[5]), trip, tuple(map(float, out[10:13]))

    def weights_m2_lse_dev(self, ll, lp, lq, beta0: float, beta: float, m: float, mean_u: float, shift: float, mp: float,
                           out: torch.Tensor):
        """weights_m2_lse with the two sums written to out[0:2] on the device (no host synchronisation)."""
        self._chk3(ll, lp, lq)
        assert out.dtype == torch.float64 and out.numel() >= 2
        check(self.lib.asmc_weights_m2_lse_dev(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, beta, m,
                                               mean_u, shift, mp, _dptr(out), self._stream), "asmc_weights_m2_lse_dev")

    # sharded importance step with the scalars left on the device (include/asmc.h asmc_weights_m2_lse_shard ...)
    def find_beta_shard_round(self, ll, lp, lq, beta0: float, target_eff: float, tol: float, world: int, n_global: int, rnd: int,
                              recs_prev: torch.Tensor | None, rec: torch.Tensor):
        """Round `rnd` of the sharded search with the previous round's decide half folded in (recs_prev = the all-gathered
        records of round rnd - 1; None in round 0): one launch per round; the caller all-gathers `rec` afterwards."""
        self._chk3(ll, lp, lq)
        assert rec.dtype == torch.float64 and rec.numel() >= _lib.ASMC_BIS_REC and rec.is_contiguous()
        assert rnd == 0 or (recs_prev is not None and recs_prev.numel() == world * _lib.ASMC_BIS_REC and recs_prev.is_contiguous())
        check(self.lib.asmc_find_beta_shard_round(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, target_eff, tol,
                                                  world, n_global, rnd, _dptr(recs_prev) if rnd else None, _dptr(rec),
                                                  self._stream), "asmc_find_beta_shard_round")

    def weights_m2_lse_shard(self, ll, lp, lq, out: torch.Tensor, recs_last: torch.Tensor | None = None, world: int = 1,
                             n_global: int = 0, beta0: float = 0.0, target_eff: float = 0.5, tol: float = 1e-6, n_rounds: int = 0):
        """`weights_m2_lse_dev` at the beta the sharded search has left on the device (no host value involved).  `recs_last`: the
        all-gathered records of the search's last round (`find_beta_shard_round` x n_rounds), closed inside this pass."""
        self._chk3(ll, lp, lq)
        assert out.dtype == torch.float64 and out.numel() >= 2
        check(self.lib.asmc_weights_m2_lse_shard(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), _dptr(out),
                                                 _dptr(recs_last), world, n_global, beta0, target_eff, tol, n_rounds, self._stream),
              "asmc_weights_m2_lse_shard")

    def normalized_weights_shard(self, ll, lp, lq, parts: torch.Tensor, world: int, rank: int, carry_uniform: float,
                                 state_copy: torch.Tensor | None = None):
        """(w, carry, tile_sums): the normalised weights from the all-gathered (m2, S1') pairs, this rank's approximate
        incoming cdf sum (a one-element device tensor) and the weights' sums per scan tile, both for `cdf_shard_records`.
        `state_copy` (40 doubles): receives the search state, for `shard_step_result`'s single read-back."""
        self._chk3(ll, lp, lq)
        assert parts.dtype == torch.float64 and parts.numel() == 2 * world and parts.is_contiguous()
        assert state_copy is None or (state_copy.dtype == torch.float64 and state_copy.numel() >= 40 and state_copy.is_contiguous())
        w, carry = torch.empty_like(ll), self.empty(1)
        tile_sums = self.empty(int(self.lib.asmc_cdf_shard_tiles(ll.numel())))
        # the gather's (ll, lp, lq, 0) records ride along where the gather would pack them itself (asmc_gather's threshold)
        pack = ll.numel() >= (1 << 16)
        check(self.lib.asmc_normalized_weights_shard(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), _dptr(parts), world,
                                                     rank, float(carry_uniform), _dptr(w), _dptr(carry), _dptr(tile_sums),
                                                     _dptr(state_copy), int(pack), self._stream), "asmc_normalized_weights_shard")
        self.rec_token = int(self.lib.asmc_rec_token(self._ctx)) if pack else 0
        return w, carry, tile_sums

    def rec_claim(self, token: int, ll, lp, lq) -> bool:
        """Directly in front of `gather(idx, x, ll, lp, lq)`: use the records `normalized_weights_shard` packed (its `rec_token`)
        if nothing has rewritten them since; False: the gather packs for itself."""
        if not token:
            return False
        return bool(self.lib.asmc_rec_claim(self._ctx, int(token), ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq)))

    def shard_step_result(self, res: torch.Tensor, world: int):
        """The sharded step's one synchronisation; res = {state copy [40], parts [2 world], info [2 world] int64 bits} in one
        device buffer.  Returns (`find_beta_shard_result` tuple, parts[world, 2], info[world, 2] int64)."""
        assert res.dtype == torch.float64 and res.numel() == 40 + 4 * world and res.is_contiguous()
        out = np.zeros(13 + 4 * world)
        check(self.lib.asmc_shard_step_result(self._ctx, _dptr(res), world, _f64p(out), self._stream), "asmc_shard_step_result")
        trip = (float(out[6]), float(out[7]), float(out[8])) if out[9] != 0.0 else None
        search = (float(out[0]), float(out[4]), bool(out[2]), int(out[3]), int(out[5]), trip, tuple(map(float, out[10:13])))
        return search, out[13:13 + 2 * world].reshape(world, 2).copy(), out[13 + 2 * world:].reshape(world, 2).astype(np.int64)

    def shard_step_finish(self, res: torch.Tensor, world: int, rank: int, cdf: torch.Tensor, kept: torch.Tensor, cap: int, x, ll, lp,
                          lq, rec_token: int = 0):
        """`shard_step_result` and - when the step read back is a finished one - this rank's search and gather enqueued from C
        right behind the synchronisation (include/asmc.h asmc_shard_step_finish), into buffers of `cap` rows allocated before
        the wait.  Returns (search tuple, parts[world, 2], info[world, 2], rows or None); rows = (x, ll, lp, lq) of this rank's
        offspring (views of the cap-row buffers)."""
        assert res.dtype == torch.float64 and res.numel() == 40 + 4 * world and res.is_contiguous()
        assert x.is_contiguous() and cdf.is_contiguous() and kept.is_contiguous() and kept.numel() >= cap
        self._chk3(ll, lp, lq)
        d = x.shape[1]
        idx = torch.empty(cap, dtype=torch.int64, device=self.device)
        xo = torch.empty((cap, d), dtype=x.dtype, device=self.device)
        llo, lpo, lqo = (torch.empty(cap, dtype=torch.float64, device=self.device) for _ in range(3))
        out = np.zeros(13 + 4 * world)
        launched = ctypes.c_int(0)
        check(self.lib.asmc_shard_step_finish(self._ctx, _dptr(res), world, rank, ll.numel(), _dptr(cdf), _dptr(kept), int(cap), _dptr(idx),
                                              d, self._xdt(x), _dptr(x), _dptr(xo), _dptr(ll), _dptr(lp), _dptr(lq), _dptr(llo), _dptr(lpo),
                                              _dptr(lqo), int(rec_token), _f64p(out), ctypes.byref(launched), self._stream),
              "asmc_shard_step_finish")
        trip = (float(out[6]), float(out[7]), float(out[8])) if out[9] != 0.0 else None
        search = (float(out[0]), float(out[4]), bool(out[2]), int(out[3]), int(out[5]), trip, tuple(map(float, out[10:13])))
        parts = out[13:13 + 2 * world].reshape(world, 2).copy()
        info = out[13 + 2 * world:].reshape(world, 2).astype(np.int64)
        rows = None
        if launched.value:
            cnt = int(info[rank, 0])
            rows = (xo[:cnt], llo[:cnt], lpo[:cnt], lqo[:cnt])
        return search, parts, info, rows

    def cdf_shard_finish_select(self, w, cdf, recs_all, tile0: int, work, state, u: torch.Tensor):
        """`cdf_shard_finish` + `select_range_dev` in three launches: (edges {fail, total, lo, hi}, buffer with the kept draws
        in front, int64 device tensor {kept, fail}).  No synchronisation."""
        assert u.dtype == torch.float64 and u.is_contiguous()
        edges, out = self.empty(4), torch.empty_like(u)
        info = torch.empty(2, dtype=torch.int64, device=self.device)
        check(self.lib.asmc_cdf_shard_finish_select(self._ctx, w.numel(), _dptr(w), _dptr(cdf), _dptr(recs_all),
                                                    int(recs_all.shape[0]), int(tile0), _dptr(work), _dptr(state), u.numel(),
                                                    _dptr(u), _dptr(edges), _dptr(out), _dptr(info), self._stream),
              "asmc_cdf_shard_finish_select")
        return edges, out, info

    def cdf_total_dev(self, out: torch.Tensor):
        """Total of the last `cdf` call -> out[0] (device to device)."""
        check(self.lib.asmc_cdf_total_dev(self._ctx, _dptr(out), self._stream), "asmc_cdf_total_dev")

    def pcg64_select(self, state4: np.ndarray, n_total: int, lo: float, hi: float) -> torch.Tensor:
        """The draws u of the next n_total PCG64 doubles with lo <= u < hi, as q = (u - lo)/(hi - lo), in the fixed
        (wave, iteration, lane) order of include/asmc.h."""
        st = np.ascontiguousarray(state4, dtype=np.uint64)
        ns = self.lib.asmc_pcg64_select_stage_len(n_total)
        if getattr(self, "_select_stage", None) is None or self._select_stage.numel() < ns:
            self._select_stage = self.empty(ns)
        cnt = ctypes.c_int64(0)
        check(self.lib.asmc_pcg64_select(self._ctx, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n_total, lo, hi,
                                         _dptr(self._select_stage), ctypes.byref(cnt), self._stream), "asmc_pcg64_select")
        q = self.empty(int(cnt.value))
        if cnt.value > 0:
            check(self.lib.asmc_pcg64_select_compact(self._ctx, n_total, _dptr(self._select_stage), _dptr(q), self._stream),
                  "asmc_pcg64_select_compact")
        return q

    def weights_m2(self, ll, lp, lq, beta0: float, beta: float, m: float, mean_u: float) -> float:
        self._chk3(ll, lp, lq)
        out = ctypes.c_double(0.0)
        check(self.lib.asmc_weights_m2(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, beta, m, mean_u,
                                       ctypes.byref(out), self._stream), "asmc_weights_m2")
        return out.value

    def weights_m2_lse(self, ll, lp, lq, beta0: float, beta: float, m: float, mean_u: float, shift: float,
                       mp: float) -> tuple[float, float]:
        """(sum (exp(lw - m) - mean_u)^2, sum exp((lw + shift) - mp)) in one pass."""
        self._chk3(ll, lp, lq)
        out = np.zeros(2)
        check(self.lib.asmc_weights_m2_lse(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, beta, m, mean_u,
                                           shift, mp, _f64p(out), self._stream), "asmc_weights_m2_lse")
        return float(out[0]), float(out[1])

    def log_weights(self, ll, lp, lq, beta0: float, beta: float, shift: float) -> torch.Tensor:
        self._chk3(ll, lp, lq)
        out = torch.empty_like(ll)
        check(self.lib.asmc_log_weights(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, beta, shift,
                                        _dptr(out), self._stream), "asmc_log_weights")
        return out

    def normalized_weights(self, ll, lp, lq, beta0: float, beta: float, shift: float, lse: float) -> torch.Tensor:
        self._chk3(ll, lp, lq)
        out = torch.empty_like(ll)
        check(self.lib.asmc_normalized_weights(self._ctx, ll.numel(), _dptr(ll), _dptr(lp), _dptr(lq), beta0, beta,
                                               shift, lse, _dptr(out), self._stream), "asmc_normalized_weights")
        return out

    def count_nonfinite(self, v: torch.Tensor) -> tuple[int, int]:
        assert v.dtype == torch.float64 and v.is_contiguous()
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self.lib.asmc_count_nonfinite(self._ctx, v.numel(), _dptr(v), ctypes.byref(a), ctypes.byref(b),
                                            self._stream), "asmc_count_nonfinite")
        return int(a.value), int(b.value)

    # ---- resampling ------------------------------------------------------------------------
    def cdf(self, w: torch.Tensor, mode: str = "exact", carry_in: float = 0.0, want_total: bool = True,
            normalize: bool = False):
        """(cdf, last element).  want_total=False leaves the total on the device (None is returned) so that
        `cdf_normalize_last` can follow without a host round trip; normalize=True writes cdf / cdf[-1] directly
        (the returned total is still the unnormalised last element)."""
        assert w.dtype == torch.float64 and w.is_contiguous()
        out = torch.empty_like(w)
        total = ctypes.c_double(0.0)
        check(self.lib.asmc_cdf(self._ctx, w.numel(), _dptr(w), _dptr(out), CDF_MODES[mode] | (_lib.ASMC_CDF_NORMALIZE if normalize else 0),
                                carry_in, ctypes.byref(total) if want_total else None, self._stream), "asmc_cdf")
        return out, (total.value if want_total else None)

    # ---- sharded exact cdf: this rank's slice of the GLOBAL sequential cumsum (include/asmc.h asmc_cdf_shard_*) ------
    def cdf_shard_records(self, w: torch.Tensor, approx_carry, first_rank: bool, tile_sums: torch.Tensor | None = None):
        """(cdf buffer, tile records [n_tiles, ASMC_CDF_REC] int64): the passes that need only an approximate
        incoming sum - a float, or a one-element device tensor when it is itself a result still on the stream
        (`normalized_weights_shard`).  No synchronisation."""
        assert w.dtype == torch.float64 and w.is_contiguous()
        n_tiles = int(self.lib.asmc_cdf_shard_tiles(w.numel()))
        cdf = torch.empty_like(w)
        rec = torch.empty((n_tiles, _lib.ASMC_CDF_REC), dtype=torch.int64, device=self.device)
        if isinstance(approx_carry, torch.Tensor):
            assert approx_carry.dtype == torch.float64 and approx_carry.numel() >= 1 and approx_carry.is_cuda
            assert tile_sums is None or (tile_sums.dtype == torch.float64 and tile_sums.numel() == n_tiles)
            check(self.lib.asmc_cdf_shard_records_dev(self._ctx, w.numel(), _dptr(w), _dptr(cdf), _dptr(approx_carry),
                                                      _dptr(tile_sums), int(first_rank), _dptr(rec), self._stream),
                  "asmc_cdf_shard_records_dev")
            return cdf, rec
        check(self.lib.asmc_cdf_shard_records(self._ctx, w.numel(), _dptr(w), _dptr(cdf), float(approx_carry), int(first_rank),
                                              _dptr(rec), self._stream), "asmc_cdf_shard_records")
        return cdf, rec

    def cdf_shard_chain(self, w: torch.Tensor, cdf: torch.Tensor, recs_all: torch.Tensor, tile0: int, work: torch.Tensor | None,
                        states_all: torch.Tensor | None, world: int, rank: int):
        """One round of the verifying chain over ALL ranks' tile records; returns (work, state[ASMC_CDF_STATE]).
        Round 1: work = states_all = None.  Round 2: the work buffer of round 1 and the all-gathered states.
        No synchronisation."""
        assert recs_all.dtype == torch.int64 and recs_all.is_contiguous() and recs_all.shape[1] == _lib.ASMC_CDF_REC
        n_total = int(recs_all.shape[0])
        if work is None:
            work = self.empty(3 * n_total)
        state = torch.empty(_lib.ASMC_CDF_STATE, dtype=torch.float64, device=self.device)  # (round 1 zeroes it itself)
        if states_all is not None:
            assert states_all.dtype == torch.float64 and states_all.is_contiguous() and states_all.numel() == world * _lib.ASMC_CDF_STATE
        check(self.lib.asmc_cdf_shard_chain(self._ctx, w.numel(), _dptr(w), _dptr(cdf), _dptr(recs_all), n_total, int(tile0),
                                            _dptr(work), _dptr(states_all), int(world), int(rank), _dptr(state), self._stream),
              "asmc_cdf_shard_chain")
        return work, state

    def cdf_shard_finish(self, w: torch.Tensor, cdf: torch.Tensor, recs_all: torch.Tensor, tile0: int, work: torch.Tensor,
                         state: torch.Tensor) -> torch.Tensor:
        """Write this rank's slice (divided by the global total of its final chain state) into `cdf`; returns the device
        vector {fail, total, lo, hi}.  No synchronisation."""
        out = self.empty(4)
        check(self.lib.asmc_cdf_shard_finish(self._ctx, w.numel(), _dptr(w), _dptr(cdf), _dptr(recs_all), int(recs_all.shape[0]),
                                             int(tile0), _dptr(work), _dptr(state), _dptr(out), self._stream),
              "asmc_cdf_shard_finish")
        return out

    def select_range(self, u: torch.Tensor, lohi: torch.Tensor) -> torch.Tensor:
        """u[(lo <= u) & (u < hi)] in index order; lohi = device tensor {lo, hi}.  Synchronises (the count)."""
        assert u.dtype == torch.float64 and u.is_contiguous() and lohi.dtype == torch.float64 and lohi.numel() >= 2
        out = torch.empty_like(u)
        cnt = ctypes.c_int64(0)
        check(self.lib.asmc_select_range(self._ctx, u.numel(), _dptr(u), _dptr(lohi), _dptr(out), ctypes.byref(cnt),
                                         self._stream), "asmc_select_range")
        return out[: int(cnt.value)]

    def select_range_dev(self, u: torch.Tensor, edges: torch.Tensor):
        """`select_range(u, edges[2:4])` enqueued only: (out buffer of u's length, int64 device tensor {kept, edges[0] as the
        failure flag}); the caller slices the buffer once it has read the count."""
        assert u.dtype == torch.float64 and u.is_contiguous() and edges.dtype == torch.float64 and edges.numel() >= 4
        out = torch.empty_like(u)
        info = torch.empty(2, dtype=torch.int64, device=self.device)
        check(self.lib.asmc_select_range_dev(self._ctx, u.numel(), _dptr(u), _dptr(edges), _dptr(out), _dptr(info),
                                             self._stream), "asmc_select_range_dev")
        return out, info

    def cdf_normalize_last(self, cdf: torch.Tensor) -> torch.Tensor:
        check(self.lib.asmc_cdf_normalize_last(self._ctx, cdf.numel(), _dptr(cdf), self._stream), "asmc_cdf_normalize_last")
        return cdf

    def cdf_normalize(self, cdf: torch.Tensor, last: float) -> torch.Tensor:
        check(self.lib.asmc_cdf_normalize(self._ctx, cdf.numel(), _dptr(cdf), last, self._stream), "asmc_cdf_normalize")
        return cdf

    def uniforms_pcg64(self, state4: np.ndarray, offset: int, n: int) -> torch.Tensor:
        st = np.ascontiguousarray(state4, dtype=np.uint64)
        out = self.empty(n)
        check(self.lib.asmc_pcg64_uniforms(self._ctx, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), offset, n,
                                           _dptr(out), self._stream), "asmc_pcg64_uniforms")
        return out

    def systematic_uniforms(self, n_out: int, j0: int, n_total: int, u0: float, v: torch.Tensor | None = None):
        out = self.empty(n_out)
        check(self.lib.asmc_systematic_uniforms(self._ctx, n_out, j0, n_total, u0, _dptr(v), _dptr(out), self._stream),
              "asmc_systematic_uniforms")
        return out

    def search(self, cdf: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
        assert cdf.dtype == torch.float64 and u.dtype == torch.float64 and cdf.is_contiguous() and u.is_contiguous()
        idx = torch.empty(u.numel(), dtype=torch.int64, device=self.device)
        check(self.lib.asmc_search(self._ctx, cdf.numel(), _dptr(cdf), u.numel(), _dptr(u), _dptr(idx), self._stream),
              "asmc_search")
        return idx

    def gather(self, idx, x, ll, lp, lq, out=None):
        """Rows `idx` of (x, ll, lp, lq); `out` = caller-owned (x, ll, lp, lq) destination buffers (no allocation)."""
        assert idx.dtype == torch.int64 and idx.is_contiguous() and x.is_contiguous()
        self._chk3(ll, lp, lq)
        n_out, d = idx.numel(), x.shape[1]
        if out is not None:
            xo, llo, lpo, lqo = out
            assert xo.shape == (n_out, d) and xo.dtype == x.dtype and xo.is_contiguous()
            self._chk3(llo, lpo, lqo)
            assert llo.numel() == n_out and lpo.numel() == n_out and lqo.numel() == n_out
        else:
            xo = torch.empty((n_out, d), dtype=x.dtype, device=self.device)
            llo, lpo, lqo = (torch.empty(n_out, dtype=torch.float64, device=self.device) for _ in range(3))
        check(self.lib.asmc_gather(self._ctx, x.shape[0], n_out, _dptr(idx), d, self._xdt(x), _dptr(x), _dptr(xo), _dptr(ll),
                                   _dptr(lp), _dptr(lq), _dptr(llo), _dptr(lpo), _dptr(lqo), self._stream), "asmc_gather")
        return xo, llo, lpo, lqo

    # ---- proposal / densities / filter -----------------------------------------------------
    def make_mixture(self, logw, mu, prec) -> DeviceMixture:
        mu = np.atleast_2d(np.asarray(mu, dtype=np.float64))
        prec = np.atleast_2d(np.asarray(prec, dtype=np.float64))
        logw = np.atleast_1d(np.asarray(logw, dtype=np.float64))
        assert mu.shape == prec.shape and logw.shape == (mu.shape[0],)
        return DeviceMixture(self.asarray(logw), self.asarray(mu), self.asarray(prec))

    def gaussian_draw(self, n, d, x_dtype, mu, sigma, seed, gid0, draw_id, want_lq=True):
        x = torch.empty((n, d), dtype=x_dtype, device=self.device)
        lq = self.empty(n) if want_lq else None
        check(self.lib.asmc_gaussian_draw(self._ctx, n, d, self._xdt(x), _dptr(mu), _dptr(sigma), seed, gid0, draw_id,
                                          _dptr(x), _dptr(lq), self._stream), "asmc_gaussian_draw")
        return x, lq

    def mixture_logpdf(self, x: torch.Tensor, mix: DeviceMixture) -> torch.Tensor:
        assert x.is_contiguous() and x.dim() == 2
        out = self.empty(x.shape[0])
        cs = mix.c_struct()
        check(self.lib.asmc_mixture_logpdf(self._ctx, x.shape[0], x.shape[1], self._xdt(x), _dptr(x), ctypes.byref(cs),
                                           _dptr(out), self._stream), "asmc_mixture_logpdf")
        return out

    def mixture_logpdf_premap(self, x: torch.Tensor, premap: torch.Tensor, mix: DeviceMixture) -> torch.Tensor:
        """log mixture(t) + sum_j h_j t_j^2 at t = clip(a x + b, lo, hi); premap = rows (a, b, lo, hi, h) of d doubles
        (include/asmc.h asmc_mixture_logpdf_premap)."""
        assert x.is_contiguous() and x.dim() == 2 and premap.dtype == torch.float64 and premap.numel() == 5 * x.shape[1]
        out = self.empty(x.shape[0])
        cs = mix.c_struct()
        check(self.lib.asmc_mixture_logpdf_premap(self._ctx, x.shape[0], x.shape[1], self._xdt(x), _dptr(x), _dptr(premap),
                                                  ctypes.byref(cs), _dptr(out), self._stream), "asmc_mixture_logpdf_premap")
        return out

    def make_coupling(self, dims: int, hidden: int, weights, biases, loc, scale) -> DeviceCoupling:
        packed = pack_coupling(self.lib, dims, hidden, weights, biases)
        scale = np.asarray(scale, dtype=np.float32)
        return DeviceCoupling(dims, len(weights) // 3, hidden, self.asarray(packed, dtype=torch.float32),
                              self.asarray(np.asarray(loc, dtype=np.float32), dtype=torch.float32),
                              self.asarray(scale, dtype=torch.float32), float(np.log(scale.astype(np.float64)).sum()))

    def make_maf(self, dims: int, hidden: int, weights, biases, loc, scale, affine: int = 0) -> DeviceCoupling:
        """A masked autoregressive flow on the device (include/asmc.h ASMC_FLOW_MAF): `weights` are the MASKED matrices."""
        packed = pack_maf(self.lib, dims, hidden, weights, biases)
        scale = np.asarray(scale, dtype=np.float32)
        return DeviceCoupling(dims, len(weights) // 3, hidden, self.asarray(packed, dtype=torch.float32),
                              self.asarray(np.asarray(loc, dtype=np.float32), dtype=torch.float32),
                              self.asarray(scale, dtype=torch.float32), float(np.log(scale.astype(np.float64)).sum()),
                              kind=_lib.ASMC_FLOW_MAF, affine=int(affine))

    def coupling_logprob(self, x: torch.Tensor, flow: DeviceCoupling) -> torch.Tensor:
        assert x.is_contiguous() and x.dim() == 2 and x.shape[1] == flow.dims
        out = self.empty(x.shape[0])
        cs = flow.c_struct()
        check(self.lib.asmc_coupling_logprob(self._ctx, x.shape[0], self._xdt(x), _dptr(x), ctypes.byref(cs), _dptr(out),
                                             self._stream), "asmc_coupling_logprob")
        return out

    def coupling_sample(self, n: int, dtype, flow: DeviceCoupling, seed: int, gid0: int, draw_id: int):
        """(x [n, dims] in `dtype`, log q(x)): n draws from the coupling flow (include/asmc.h asmc_coupling_sample)."""
        x = torch.empty((n, flow.dims), dtype=dtype, device=self.device)
        lq = self.empty(n)
        cs = flow.c_struct()
        check(self.lib.asmc_coupling_sample(self._ctx, n, self._xdt(x), ctypes.byref(cs), int(seed) & (2**64 - 1), int(gid0),
                                            int(draw_id) & 0xFFFFFFFF, _dptr(x), _dptr(lq), self._stream), "asmc_coupling_sample")
        return x, lq

    def make_transform(self, kind, periodic, lower, upper, mean=None, std=None, eps=1e-6, unit_logj=0.0,
                       affine_logj=0.0) -> DeviceTransform:
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32), device=self.device)  # noqa: E731
        k, per = np.asarray(kind), np.asarray(periodic)
        hints = (1 if not per.any() else 0) | (2 if not (k == 1).any() else 0) | (4 if not (k == 2).any() else 0)
        return DeviceTransform(len(kind), i32(kind), i32(periodic), self.asarray(np.asarray(lower, dtype=np.float64)),
                               self.asarray(np.asarray(upper, dtype=np.float64)),
                               None if mean is None else self.asarray(np.asarray(mean, dtype=np.float64)),
                               None if std is None else self.asarray(np.asarray(std, dtype=np.float64)), float(eps),
                               float(unit_logj), float(affine_logj), int(hints))

    def _transform(self, fn, name, x: torch.Tensor, t: DeviceTransform, want_logj: bool):
        assert x.is_contiguous() and x.dim() == 2 and x.shape[1] == t.d
        out = torch.empty_like(x)
        lj = self.empty(x.shape[0]) if want_logj else None
        cs = t.c_struct()
        check(fn(self._ctx, x.shape[0], self._xdt(x), _dptr(x), _dptr(out), _dptr(lj), ctypes.byref(cs), self._stream), name)
        return out, lj

    def transform_forward(self, x: torch.Tensor, t: DeviceTransform, want_logj: bool = True):
        """(z, log|det J|) of x -> z."""
        return self._transform(self.lib.asmc_transform_forward, "asmc_transform_forward", x, t, want_logj)

    def transform_inverse(self, z: torch.Tensor, t: DeviceTransform, want_logj: bool = True):
        """(x, log|det J|) of z -> x."""
        return self._transform(self.lib.asmc_transform_inverse, "asmc_transform_inverse", z, t, want_logj)

    def compact_valid(self, x, ll, lp, lq):
        self._chk3(ll, lp, lq)
        n, d = x.shape
        xo = torch.empty_like(x)
        llo, lpo, lqo = torch.empty_like(ll), torch.empty_like(lp), torch.empty_like(lq)
        nv = ctypes.c_int64(0)
        check(self.lib.asmc_compact_valid(self._ctx, n, d, self._xdt(x), _dptr(x), _dptr(ll), _dptr(lp), _dptr(lq),
                                          _dptr(xo), _dptr(llo), _dptr(lpo), _dptr(lqo), ctypes.byref(nv), self._stream),
              "asmc_compact_valid")
        k = int(nv.value)
        if k == n:  # every row valid: the library copied nothing (include/asmc.h)
            return x, ll, lp, lq
        return xo[:k], llo[:k], lpo[:k], lqo[:k]

    # ---- moments / pCN ---------------------------------------------------------------------
    def colsum(self, x: torch.Tensor) -> np.ndarray:
        n, d = x.shape
        out = np.empty(d)
        check(self.lib.asmc_colsum(self._ctx, n, d, self._xdt(x), _dptr(x), _f64p(out), self._stream), "asmc_colsum")
        return out

    def centered_gram(self, x: torch.Tensor, center: np.ndarray) -> np.ndarray:
        n, d = x.shape
        c = np.ascontiguousarray(center, dtype=np.float64)
        out = np.empty((d, d))
        check(self.lib.asmc_centered_gram(self._ctx, n, d, self._xdt(x), _dptr(x), _f64p(c), _f64p(out), self._stream),
              "asmc_centered_gram")
        return out

    def mean_gram(self, x: torch.Tensor, n_mean: int, comm=None) -> tuple[np.ndarray, np.ndarray]:
        """(column sums, Gram matrix centred on sums / n_mean) in one enqueue with one synchronisation; equal to
        colsum -> division -> centered_gram bit for bit.  With a sharded `comm` that has an RCCL communicator for the
        library (`use_rccl`), both are summed over the ranks on the stream (n_mean: the global population)."""
        n, d = x.shape
        s, g = np.empty(d), np.empty((d, d))
        across = int(comm is not None and comm.sharded)
        if across and x.data_ptr() % 16:
            x = x.clone()  # the matrix-core Gram kernel reads 16-byte pieces (the call synchronises: the copy lives long enough)
        check(self.lib.asmc_mean_gram(self._ctx, n, d, self._xdt(x), _dptr(x), int(n_mean), across, _f64p(s), _f64p(g),
                                      self._stream), "asmc_mean_gram")
        return s, g

    def mean_gram_enqueue(self, x: torch.Tensor, n_mean: int, comm=None, gathered: bool = False) -> bool:
        """Start `mean_gram` on the stream without waiting; False for shapes without the device-side path (and, for a sharded
        `comm`, without a communicator for the library's own all-reduces: `use_rccl`)."""
        n, d = x.shape
        across = int(comm is not None and comm.sharded)
        if not (d in (32, 64, 128) and not os.environ.get("ASMC_GRAM_GENERIC")):
            return False
        if across and not self.use_rccl(comm):
            return False
        if x.data_ptr() % 16:
            # a rank-local property must not pick the code path of a sharded run (the other ranks would wait in a collective
            # this rank never issues): single rank - the caller's fallback; sharded - an aligned copy, kept until the fetch
            if not across:
                return False
            x = self._gram_keep = x.clone()
        # gathered: x is what the last `gather` returned and nothing has rewritten it since - its column sums rode along the gather
        check(self.lib.asmc_mean_gram_enqueue(self._ctx, n, d, self._xdt(x), _dptr(x), int(n_mean), across | (2 if gathered else 0),
                                              self._stream), "asmc_mean_gram_enqueue")
        self._gram_gen = getattr(self, "_gram_gen", 0) + 1  # names the request: a fetch is for the LATEST one only
        return True

    def mean_gram_fetch(self, d: int) -> tuple[np.ndarray, np.ndarray]:
        s, g = np.empty(d), np.empty((d, d))
        check(self.lib.asmc_mean_gram_fetch(self._ctx, d, _f64p(s), _f64p(g), self._stream), "asmc_mean_gram_fetch")
        return s, g

    def colsum_dev(self, x: torch.Tensor, gathered: bool = False) -> torch.Tensor:
        """Column sums of this rank's rows as a device tensor (no synchronisation): the caller all-reduces them on the stream."""
        n, d = x.shape
        s = self.empty(d)
        check(self.lib.asmc_colsum_dev(self._ctx, n, d, self._xdt(x), _dptr(x), int(bool(gathered)), _dptr(s), self._stream),
              "asmc_colsum_dev")
        return s

    def centered_gram_dev(self, x: torch.Tensor, sums: torch.Tensor, n_mean: int) -> torch.Tensor:
        """Gram matrix of this rank's rows around sums / n_mean (device tensors in and out, no synchronisation); d in {32, 64, 128}."""
        n, d = x.shape
        g = self.empty((d, d))
        check(self.lib.asmc_centered_gram_dev(self._ctx, n, d, self._xdt(x), _dptr(x), _dptr(sums), int(n_mean), _dptr(g),
                                              self._stream), "asmc_centered_gram_dev")
        return g

    def reference_factor(self, d: int, n_mean: int, n_cov: int, moments=None, moments_dev=None):
        """(mu, L, Linv) of the mutation's reference Gaussian as device tensors, factored ON the device behind the moments of
        the pending `mean_gram_enqueue` (moments=None: consumed, no fetch follows) or of `moments` = (sums, Gram) merged on the
        host (include/asmc.h asmc_reference_factor).  Nothing is synchronised: `reference_factor_status()` after the next
        synchronisation says whether the covariance could be factored."""
        seg = -(-d // 32) * 32
        size = seg + 2 * seg * d
        bufs = self.__dict__.setdefault("_ref_out", {})
        # two buffers in turn: the previous mutation's kernels may still read theirs when the next fit is enqueued
        slot = bufs["slot"] = 1 - bufs.get("slot", 1)
        if bufs.get(("buf", slot)) is None or bufs[("buf", slot)].numel() != size:
            bufs[("buf", slot)] = torch.zeros(size, dtype=torch.float64, device=self.device)
        out = bufs[("buf", slot)]
        if moments_dev is not None:  # (sums, Gram) already on the device (summed over the ranks by the caller)
            s_d, g_d = moments_dev
            assert s_d.dtype == torch.float64 and g_d.dtype == torch.float64 and s_d.numel() == d and g_d.numel() == d * d
            self._ref_keep = (s_d, g_d)  # the kernel reads them when the stream gets there
            check(self.lib.asmc_reference_factor_dev(self._ctx, d, int(n_mean), int(n_cov), _dptr(s_d), _dptr(g_d), _dptr(out),
                                                     self._stream), "asmc_reference_factor_dev")
            self.ref_generation = int(self.lib.asmc_reference_factor_generation(self._ctx))
            return out[:d], out[seg:seg + d * d].view(d, d), out[seg + seg * d:seg + seg * d + d * d].view(d, d)
        if moments is None:
            sp, gp = None, None
        else:
            s = np.ascontiguousarray(moments[0], dtype=np.float64)
            g = np.ascontiguousarray(moments[1], dtype=np.float64)
            assert s.shape == (d,) and g.shape == (d, d)
            sp, gp = _f64p(s), _f64p(g)
        check(self.lib.asmc_reference_factor(self._ctx, d, int(n_mean), int(n_cov), sp, gp, _dptr(out), self._stream),
              "asmc_reference_factor")
        self.ref_generation = int(self.lib.asmc_reference_factor_generation(self._ctx))  # (reference_factor_status(generation))
        return out[:d], out[seg:seg + d * d].view(d, d), out[seg + seg * d:seg + seg * d + d * d].view(d, d)

    def reference_factor_status(self, generation: int | None = None) -> int:
        """Jitter tries the last `reference_factor` needed (0: none); -1: the covariance was not factorable; -2: the stream has
        not been synchronised since.  `generation`: of that particular request (`ref_generation` right after the call) - the
        latest one may be the NEXT temperature's, still waiting on the stream."""
        st = ctypes.c_int(0)
        if generation is not None:
            check(self.lib.asmc_reference_factor_status_of(self._ctx, int(generation), ctypes.byref(st)),
                  "asmc_reference_factor_status_of")
            return int(st.value)
        check(self.lib.asmc_reference_factor_status(self._ctx, ctypes.byref(st)), "asmc_reference_factor_status")
        return int(st.value)

    def mean_gram_across_ranks_ok(self, x: torch.Tensor, comm) -> bool:
        """The shapes asmc_mean_gram sums over the ranks itself (the fp64-MFMA Gram kernel's), given a communicator."""
        # (only properties every rank shares: the alignment of this rank's rows is dealt with inside mean_gram)
        return x.shape[1] in (32, 64, 128) and not os.environ.get("ASMC_GRAM_GENERIC") and self.use_rccl(comm)

    # ---- Student-t reference fit (tpCN): per-particle half of the EM on a device-resident subsample ----------------
    def student_estep(self, xs: torch.Tensor, mu: np.ndarray, linv: np.ndarray, nu: float):
        """(z [m] on device, sum z, sum (log z - z), sum z x [d]) for the subsample xs [m, d] fp64."""
        assert xs.dtype == torch.float64 and xs.is_contiguous() and xs.dim() == 2
        m, d = xs.shape
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        linv = np.ascontiguousarray(linv, dtype=np.float64)
        z = self.empty(m)
        sums = np.empty(d + 2)
        check(self.lib.asmc_student_estep(self._ctx, m, d, _dptr(xs), _f64p(mu), _f64p(linv), float(nu), _dptr(z), _f64p(sums),
                                          self._stream), "asmc_student_estep")
        return z, float(sums[0]), float(sums[1]), sums[2:].copy()

    def student_scale(self, xs: torch.Tensor, z: torch.Tensor, mu: np.ndarray) -> torch.Tensor:
        """r = sqrt(z) (x - mu); centered_gram(r, 0) is the weighted scatter matrix of the M-step."""
        m, d = xs.shape
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        r = torch.empty_like(xs)
        check(self.lib.asmc_student_scale(self._ctx, m, d, _dptr(xs), _dptr(z), _f64p(mu), _dptr(r), self._stream),
              "asmc_student_scale")
        return r

    def student_fit(self, xs: torch.Tensor, max_iter: int, rtol: float, nu0: float):
        """The whole Student-t EM on the device (include/asmc.h asmc_student_fit): ((mu, L, Linv) device views, nu, iterations,
        status, mu_host, Sigma_host); None for shapes the device-side EM does not take (the host-driven EM serves them)."""
        assert xs.dtype == torch.float64 and xs.is_contiguous() and xs.dim() == 2
        m, d = xs.shape
        if d not in (32, 64, 128) or os.environ.get("ASMC_GRAM_GENERIC"):
            return None
        if xs.data_ptr() % 16:  # alignment is a rank-local accident: never let it choose the code path (ranks must agree bit for bit)
            xs = xs.clone()
        seg = -(-d // 32) * 32
        size = seg + 2 * seg * d
        bufs = self.__dict__.setdefault("_ref_out", {})
        slot = bufs["slot"] = 1 - bufs.get("slot", 1)
        if bufs.get(("buf", slot)) is None or bufs[("buf", slot)].numel() != size:
            bufs[("buf", slot)] = torch.zeros(size, dtype=torch.float64, device=self.device)
        out = bufs[("buf", slot)]
        scr = self.__dict__.setdefault("_student_scratch", {})
        if scr.get("shape") != (m, d):
            scr.update(shape=(m, d), r=torch.empty((m, d), dtype=torch.float64, device=self.device), z=self.empty(m))
        res = np.empty(4 + d + d * d)
        check(self.lib.asmc_student_fit(self._ctx, m, d, _dptr(xs), int(max_iter), float(rtol), float(nu0), _dptr(scr["r"]),
                                        _dptr(scr["z"]), _dptr(out), _f64p(res), self._stream), "asmc_student_fit")
        views = (out[:d], out[seg:seg + d * d].view(d, d), out[seg + seg * d:seg + seg * d + d * d].view(d, d))
        return views, float(res[0]), int(res[1]), int(res[2]), res[4:4 + d].copy(), res[4 + d:].reshape(d, d).copy()

    def use_rccl(self, comm) -> bool:
        """Hand the library the communicator `comm.rccl_direct()` makes for the collectives it issues itself (include/asmc.h
        asmc_set_rccl); False when the communicator has none (single rank, gloo rigs, ASMC_RCCL_DIRECT=0)."""
        direct = comm.rccl_direct() if comm is not None and hasattr(comm, "rccl_direct") else None
        if direct is None:
            return False
        if getattr(self, "_rccl_set", None) != direct:
            check(self.lib.asmc_set_rccl(self._ctx, ctypes.c_void_p(direct[0]), ctypes.c_void_p(direct[1])), "asmc_set_rccl")
            check(self.lib.asmc_set_rccl_allgather(self._ctx, ctypes.c_void_p(direct[2])), "asmc_set_rccl_allgather")
            self._rccl_set = direct
        return True

    def all_gather(self, comm, t: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """`comm.all_gather_tensor(t)` for the small fp64 / int64 exchanges of the sharded hot path: through the library's own
        communicator on its own stream when there is one (`use_rccl`), else through torch.distributed.  `out`: a contiguous
        buffer of world * t.numel() elements of t's dtype to gather into (a slice of a larger result buffer)."""
        assert out is None or (out.dtype == t.dtype and out.numel() == comm.world * t.numel() and out.is_contiguous())
        if t.dtype in (torch.float64, torch.int64) and t.is_contiguous() and t.is_cuda and self.use_rccl(comm):
            if out is None:
                out = torch.empty((comm.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            check(self.lib.asmc_rccl_all_gather(self._ctx, _dptr(t), _dptr(out), t.numel(), int(t.dtype == torch.int64),
                                                self._stream), "asmc_rccl_all_gather")
            return out
        if out is not None:
            return comm.all_gather_into(out, t.contiguous())
        return comm.all_gather_tensor(t)

    def set_count_hook(self, comm, n_global: int | None):
        """Sharded mutation: let `pcn_mutate` / `pcn_mutate_flow` adapt the step size from the GLOBAL acceptance rate
        with the whole step loop enqueued on the stream - after each step the library leaves this rank's accept
        count in a device cell and calls back; the callback all-reduces the cell over `comm` (RCCL, on the stream).
        `n_global=None` (or a single-rank comm) removes the hook.  Accept counts returned afterwards are global."""
        if n_global is None or comm is None or not comm.sharded:
            check(self.lib.asmc_pcn_set_count_hook(self._ctx, None, None, None, 0), "asmc_pcn_set_count_hook")
            self._hook = None
            return
        # ASMC_MAX_COUNT_CELLS cells: a lagged adaptation (adapt_lag = k) exchanges the k counts of a block at once - step t writes
        # cell t % k, one all-reduce sums them all (unused cells stay zero)
        cell = self.__dict__.get("_count_cell")  # (the library writes its cells before every exchange: one allocation, no fill per call)
        if cell is None:
            cell = self._count_cell = torch.zeros(_lib.ASMC_MAX_COUNT_CELLS, dtype=torch.int64, device=self.device)
        if self.use_rccl(comm):  # the library issues the all-reduce itself, on its own stream
            self._hook = (None, cell, None)
            check(self.lib.asmc_pcn_set_count_rccl(self._ctx, _dptr(cell), int(n_global)), "asmc_pcn_set_count_rccl")
            self._count_active = 1
            return

        def cb(_user, _stream):
            try:
                k = self.__dict__.get("_count_active", 1)
                comm.all_reduce_sum_(cell if k >= cell.numel() else cell[:k])  # (k = 1 unless the adaptation is lagged)
                return 0
            except Exception as exc:  # exceptions must not unwind through the C frames
                self._hook_error = exc
                return 1

        fn = _lib.COUNT_HOOK(cb)
        self._hook = (fn, cell, cb)  # keep the trampoline and the cell alive while installed
        check(self.lib.asmc_pcn_set_count_hook(self._ctx, ctypes.cast(fn, ctypes.c_void_p), None, _dptr(cell), int(n_global)),
              "asmc_pcn_set_count_hook")
        self._count_active = 1

    def _count_cells_for(self, adapt) -> None:
        """Before a device-side step loop of a sharded run: the exchange sums exactly the cells the loop writes - one, or k for a
        lagged adaptation (`adapt` = k >= 2) - so that the default schedule's per-step all-reduce stays an 8-byte one."""
        if self.__dict__.get("_hook") is None:
            return
        k = max(1, int(adapt))
        if k != self.__dict__.get("_count_active"):
            if k > _lib.ASMC_MAX_COUNT_CELLS:
                raise ValueError(f"adapt_lag {k} exceeds {_lib.ASMC_MAX_COUNT_CELLS}")
            check(self.lib.asmc_pcn_set_count_cells(self._ctx, k), "asmc_pcn_set_count_cells")
            self._count_active = k

    def pcn_mutate(self, x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, t_lq, seed, gid0, rho, n_steps, step0=0,
                   target_accept=0.234, adapt=True, noise="f64", nu=0.0):
        """n_steps fused pCN steps in place (nu > 0: t-preconditioned steps with a Student-t reference).
        Returns (n_accept[n_steps], rho_hist[n_steps], rho_out)."""
        self._chk3(ll, lp, lq)
        self._count_cells_for(adapt)
        n, d = x.shape
        prm = AsmcPcnParams(d, self._xdt(x), beta, mu.data_ptr(), L.data_ptr(), Linv.data_ptr(), t_ll.c_struct(),
                            t_lp.c_struct(), t_lq.c_struct(), seed, gid0, target_accept, int(adapt),
                            {"f64": 0, "f32": 1}[noise], float(nu))
        n_acc = np.zeros(n_steps, dtype=np.int64)
        rho_hist = np.zeros(n_steps)
        rho_io = ctypes.c_double(rho)
        check(self.lib.asmc_pcn_mutate(self._ctx, n, _dptr(x), _dptr(ll), _dptr(lp), _dptr(lq), ctypes.byref(prm),
                                       n_steps, step0, ctypes.byref(rho_io),
                                       n_acc.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _f64p(rho_hist),
                                       self._stream), "asmc_pcn_mutate")
        return n_acc, rho_hist, rho_io.value

    def pcn_mutate_flow(self, x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, flow: DeviceCoupling, seed, gid0, rho,
                        n_steps, step0=0, target_accept=0.234, adapt=True, noise="f64", nu=0.0):
        """pcn_mutate with a coupling-flow proposal density (log_q evaluated on the MFMA each step)."""
        self._chk3(ll, lp, lq)
        self._count_cells_for(adapt)
        n, d = x.shape
        prm = AsmcPcnParams(d, self._xdt(x), beta, mu.data_ptr(), L.data_ptr(), Linv.data_ptr(), t_ll.c_struct(),
                            t_lp.c_struct(), t_lp.c_struct(), seed, gid0, target_accept, int(adapt),
                            {"f64": 0, "f32": 1}[noise], float(nu))
        nbytes = self.lib.asmc_pcn_flow_work_bytes(n, d, self._xdt(x))
        if self._flow_work is None or self._flow_work.numel() < nbytes:
            self._flow_work = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        cs = flow.c_struct()
        n_acc = np.zeros(n_steps, dtype=np.int64)
        rho_hist = np.zeros(n_steps)
        rho_io = ctypes.c_double(rho)
        check(self.lib.asmc_pcn_mutate_flow(self._ctx, n, _dptr(x), _dptr(ll), _dptr(lp), _dptr(lq), ctypes.byref(prm),
                                            ctypes.byref(cs), _dptr(self._flow_work), nbytes, n_steps, step0,
                                            ctypes.byref(rho_io), n_acc.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                            _f64p(rho_hist), self._stream), "asmc_pcn_mutate_flow")
        return n_acc, rho_hist, rho_io.value

    def pcn_mutate_flow_enqueue(self, x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, flow: DeviceCoupling, seed, gid0, rho,
                                n_steps, step0=0, target_accept=0.234, adapt=True, noise="f64", nu=0.0):
        """`pcn_mutate_flow` without the wait: everything is on the stream when this returns, `pcn_mutate_flow_result(handle)`
        synchronises and returns what `pcn_mutate_flow` returns.  The caller may enqueue more behind it in between."""
        self._chk3(ll, lp, lq)
        self._count_cells_for(adapt)
        n, d = x.shape
        prm = AsmcPcnParams(d, self._xdt(x), beta, mu.data_ptr(), L.data_ptr(), Linv.data_ptr(), t_ll.c_struct(),
                            t_lp.c_struct(), t_lp.c_struct(), seed, gid0, target_accept, int(adapt),
                            {"f64": 0, "f32": 1}[noise], float(nu))
        nbytes = self.lib.asmc_pcn_flow_work_bytes(n, d, self._xdt(x))
        if self._flow_work is None or self._flow_work.numel() < nbytes:
            self._flow_work = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        cs = flow.c_struct()
        check(self.lib.asmc_pcn_mutate_flow_enqueue(self._ctx, n, _dptr(x), _dptr(ll), _dptr(lp), _dptr(lq), ctypes.byref(prm),
                                                    ctypes.byref(cs), _dptr(self._flow_work), nbytes, n_steps, step0,
                                                    float(rho), self._stream), "asmc_pcn_mutate_flow_enqueue")
        return (int(n_steps), (mu, L, Linv, t_ll, t_lp, flow, prm, cs))  # keeps the kernels' tables alive until the result

    def pcn_mutate_flow_result(self, handle):
        n_steps = handle[0]
        n_acc = np.zeros(n_steps, dtype=np.int64)
        rho_hist = np.zeros(n_steps)
        rho_out = ctypes.c_double(0.0)
        check(self.lib.asmc_pcn_mutate_flow_result(self._ctx, n_steps, ctypes.byref(rho_out),
                                                   n_acc.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _f64p(rho_hist),
                                                   self._stream), "asmc_pcn_mutate_flow_result")
        return n_acc, rho_hist, rho_out.value

    def pcn_lq_nan(self) -> int:
        """NaNs in the carried log q after the last pcn_mutate / pcn_mutate_flow call (counted by the call itself)."""
        return int(self.lib.asmc_pcn_lq_nan(self._ctx))

    def pcn_flow_nonfinite(self) -> int:
        """Proposals of the last `pcn_mutate_flow` whose flow density was not finite (rejected)."""
        return int(self.lib.asmc_pcn_flow_nonfinite(self._ctx))

    def pcn_propose(self, x, mu, L, Linv, rho, seed, gid0, step, nu=0.0):
        n, d = x.shape
        xp = torch.empty_like(x)
        q0, q1 = self.empty(n), self.empty(n)
        check(self.lib.asmc_pcn_propose(self._ctx, n, d, self._xdt(x), _dptr(x), _dptr(xp), _dptr(q0), _dptr(q1),
                                        _dptr(mu), _dptr(L), _dptr(Linv), rho, float(nu), seed, gid0, step, self._stream),
              "asmc_pcn_propose")
        return xp, q0, q1

    def pcn_accept(self, x, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, q0, q1, beta, seed, gid0, step,
                   logj_old=None, logj_new=None, want_count: bool = True):
        """want_count=False leaves the accept count on the device (pcn_split_adapt consumes it): no synchronisation."""
        n, d = x.shape
        nacc = ctypes.c_int64(0)
        check(self.lib.asmc_pcn_accept(self._ctx, n, d, self._xdt(x), _dptr(x), _dptr(x_prop), _dptr(ll), _dptr(lp),
                                       _dptr(lq), _dptr(ll_new), _dptr(lp_new), _dptr(lq_new), _dptr(logj_old),
                                       _dptr(logj_new), _dptr(q0), _dptr(q1), beta, seed, gid0, step,
                                       ctypes.byref(nacc) if want_count else None, self._stream), "asmc_pcn_accept")
        return int(nacc.value) if want_count else None

    # whitened-state session of the split path: y = L^-1 (x - mu) stays coordinate-major in the context between the calls
    def pcn_ysplit_begin(self, x, beta, mu, L, Linv, seed, gid0, rho, target_accept=0.234, adapt=True, nu=0.0,
                         noise="f64"):
        """Opens a session on x (left untouched until pcn_ysplit_end) and returns its handle, or None when the dimension has
        no whitened-state kernels (the caller then uses pcn_propose / pcn_accept)."""
        n, d = x.shape
        prm = AsmcPcnParams(d, self._xdt(x), beta, mu.data_ptr(), L.data_ptr(), Linv.data_ptr(), AsmcMixture(), AsmcMixture(),
                            AsmcMixture(), seed, gid0, target_accept, int(adapt), {"f64": 0, "f32": 1}[noise], float(nu))
        rc = self.lib.asmc_pcn_ysplit_begin(self._ctx, n, _dptr(x), ctypes.byref(prm), float(rho), self._stream)
        if rc == _lib.ASMC_ERR_UNSUPPORTED:
            return None
        check(rc, "asmc_pcn_ysplit_begin")
        return {"prm": prm, "x": x, "keep": (mu, L, Linv)}

    def pcn_ysplit_propose(self, sess, step: int):
        x = sess["x"]
        xp = torch.empty_like(x)
        check(self.lib.asmc_pcn_ysplit_propose(self._ctx, x.shape[0], ctypes.byref(sess["prm"]), step, _dptr(xp), self._stream),
              "asmc_pcn_ysplit_propose")
        return xp

    def pcn_ysplit_propose_tr(self, sess, step: int, t: DeviceTransform, logq=None):
        """`pcn_ysplit_propose` for a chain in the preconditioned space of `t`: returns (x', log|det dT^-1/dz| at z', log q(x')
        or None).  `logq` = (premap tensor [5 d], DeviceMixture with one component, minus_logj) evaluates the proposal
        flow's density from z' in the same pass (include/asmc.h asmc_pcn_ysplit_propose_tr)."""
        x = sess["x"]
        xp = torch.empty_like(x)
        logj = self.empty(x.shape[0])
        lq = self.empty(x.shape[0]) if logq is not None else None
        ts = t.c_struct()
        if logq is not None:
            premap, mix, minus = logq
            ms = mix.c_struct()
            check(self.lib.asmc_pcn_ysplit_propose_tr(self._ctx, x.shape[0], ctypes.byref(sess["prm"]), step, ctypes.byref(ts),
                                                      _dptr(premap), ctypes.byref(ms), int(bool(minus)), _dptr(xp), _dptr(logj),
                                                      _dptr(lq), self._stream), "asmc_pcn_ysplit_propose_tr")
        else:
            check(self.lib.asmc_pcn_ysplit_propose_tr(self._ctx, x.shape[0], ctypes.byref(sess["prm"]), step, ctypes.byref(ts),
                                                      None, None, 0, _dptr(xp), _dptr(logj), None, self._stream),
                  "asmc_pcn_ysplit_propose_tr")
        return xp, logj, lq

    def pcn_ysplit_accept(self, sess, step: int, ll, lp, lq, ll_new, lp_new, lq_new, n_global: int, t: int,
                          logj=None, logj_new=None):
        """logj / logj_new: carried / proposed log-Jacobian of a chain in a preconditioned space (both or neither)."""
        self._chk3(ll, lp, lq)
        ll_new, lp_new, lq_new = (v.to(torch.float64).contiguous() for v in (ll_new, lp_new, lq_new))
        if logj is not None:
            assert logj.dtype == torch.float64 and logj.is_contiguous()
            logj_new = logj_new.to(torch.float64).contiguous()
        check(self.lib.asmc_pcn_ysplit_accept(self._ctx, sess["x"].shape[0], ctypes.byref(sess["prm"]), step, _dptr(ll), _dptr(lp),
                                              _dptr(lq), _dptr(ll_new), _dptr(lp_new), _dptr(lq_new), _dptr(logj),
                                              _dptr(logj_new), int(n_global), int(t), self._stream), "asmc_pcn_ysplit_accept")

    def pcn_ysplit_end(self, sess, n_steps: int):
        """Writes the chain state back into the session's x; returns what pcn_split_end returns."""
        x = sess["x"]
        check(self.lib.asmc_pcn_ysplit_end(self._ctx, x.shape[0], _dptr(x), ctypes.byref(sess["prm"]), self._stream),
              "asmc_pcn_ysplit_end")
        return self.pcn_split_end(n_steps)

    # split path with the step size and the accept counts resident on the device (no host round trip per step)
    def pcn_split_begin(self, rho: float):
        check(self.lib.asmc_pcn_split_begin(self._ctx, float(rho), self._stream), "asmc_pcn_split_begin")

    def pcn_split_adapt(self, n_global: int, target_accept: float, t: int, adapt: bool = True):
        check(self.lib.asmc_pcn_split_adapt(self._ctx, int(n_global), float(target_accept), int(t), int(adapt), self._stream),
              "asmc_pcn_split_adapt")

    def pcn_split_end(self, n_steps: int):
        """(n_accept[n_steps] (global when an exchange hook is installed), rho_hist[n_steps], rho) - synchronises."""
        n_acc = np.zeros(n_steps, dtype=np.int64)
        hist = np.zeros(n_steps)
        rho = ctypes.c_double(0.0)
        check(self.lib.asmc_pcn_split_end(self._ctx, n_steps, n_acc.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _f64p(hist),
                                          ctypes.byref(rho), self._stream), "asmc_pcn_split_end")
        return n_acc, hist, rho.value
