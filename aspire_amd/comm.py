"""Particle sharding across ranks (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

The reference has no distributed code; this is new design.  Rank r owns the
contiguous block [r*N/G, (r+1)*N/G) of the GLOBAL particle order, so a 1-rank and a G-rank run with
the same generator produce the same global resample indices.

Collectives used by the hot path:
  C1  all-reduce(max) + all-gather of the per-rank (S1, S2) sums per candidate beta; merged on every
      rank in rank order (bitwise identical on all ranks)
  C2  all-gather of the normalised weights (8 B/particle); every rank then runs the same exact
      (sequential-order) scan over all N, so no rank-to-rank dependency chain exists
  C3  all-to-all of requested rows (index lists, then rows)
  C4  all-gather of column sums / centred Gram partials, accept counts
Scalars travel as small fp64 tensors; with xGMI's point-to-point links an all-gather of 24*K bytes
is latency bound, so every scalar exchange is ONE all-gather followed by a local rank-ordered merge.
"""
from __future__ import annotations

import os

import numpy as np
import torch


class Comm:
    """Single-process communicator (world size 1)."""

    rank = 0
    world = 1
    force_sharded = False  # test rig: take the sharded code path with a one-rank group (tools/rig1.sh)

    @property
    def sharded(self) -> bool:
        """True when the population is split over ranks, i.e. when the collective forms of the hot path run."""
        return self.world > 1 or self.force_sharded

    def all_gather_f64(self, arr) -> np.ndarray:
        return np.asarray(arr, dtype=np.float64)[None, ...]

    def all_reduce_max_f64(self, arr) -> np.ndarray:
        return np.asarray(arr, dtype=np.float64)

    def all_gather_tensor(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_to_all_rows(self, send: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
        return send

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """all_gather_tensor into a caller-owned buffer (hot loops: no allocation per call)."""
        out.copy_(t)
        return out

    def all_gather_ragged(self, t: torch.Tensor, counts) -> torch.Tensor:
        """Concatenate shards of DIFFERENT lengths along dim 0 in rank order; counts[r] = rows of rank r's shard, the
        same list on every rank."""
        return t

    def all_gather_i64(self, arr) -> np.ndarray:
        return np.asarray(arr, dtype=np.int64)[None, ...]

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def barrier(self):
        pass

    def signal(self, tag: str, error: BaseException | str | None = None) -> None:
        """One rank tells the others that a long host-side job (flow training) is done, or failed - see TorchDistComm.signal."""

    def await_signal(self, tag: str, timeout_s: float = 86400.0) -> None:
        """Wait for `signal(tag)` of the producing rank without sitting in a collective."""

    def chain_recv(self) -> float | None:
        return None

    def chain_send(self, value: float) -> None:
        pass

    def broadcast_f64(self, value: float, src: int) -> float:
        return value


class TorchDistComm(Comm):
    """torch.distributed communicator; tensors live on `device` (cuda for nccl/RCCL, cpu for gloo)."""

    def __init__(self, device: torch.device | str, group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        # gloo moves host memory only: device tensors are staged through the host (debugging / CPU-only test
        # rigs; the production backend is "nccl" = RCCL, which exchanges device memory over xGMI directly)
        self._stage = dist.get_backend(group) == "gloo"

    def rccl_direct(self):
        """(address of ncclAllReduce, ncclComm_t) of a communicator of this group's ranks that belongs to the HIP library's
        own launches (include/asmc.h asmc_pcn_set_count_rccl): the per-step accept-count exchange is then an RCCL kernel on
        the step kernels' stream - torch.distributed runs its collectives on a stream of its own, and every call costs
        two stream hand-overs (about 13 us of idle GPU per step boundary, profiles/r02_rig1_*).  The communicator is made
        the textbook way - rank 0 draws an ncclUniqueId, the group broadcasts it, every rank calls ncclCommInitRank - on
        the RCCL that torch has loaded.  None when the backend is not RCCL (gloo rigs) or it is switched off.

        Default (ASMC_RCCL_DIRECT unset): ON for a one-rank group - the rig every collective of this path has been run and
        checked on (tools/nccl_world1.py) - and OFF for world > 1, where the collectives go through torch.distributed: no
        multi-GPU node has been available to validate the library's own communicator with real peers, and a
        mismatch there is a hang, not an error.  ASMC_RCCL_DIRECT=1 turns it on for any world size, =0 off.
        The set-up is itself a collective: `HipSMC.sample` calls it once on every rank right after `sync_rng`, so that no
        rank-local condition later in the run decides whether a rank takes part in it."""
        if hasattr(self, "_rccl"):
            return self._rccl
        self._rccl = None
        want = os.environ.get("ASMC_RCCL_DIRECT", "1" if self.world == 1 else "0")
        if self._stage or self.device.type != "cuda" or want == "0":
            return None
        import ctypes
        import logging

        log = logging.getLogger(__name__)

        def all_agree(ok: bool) -> bool:
            """Every step below is taken by all ranks or by none: the ranks agree on each outcome through the group."""
            flag = torch.tensor([0 if ok else 1], dtype=torch.int64, device=self.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.SUM, group=self.group)
            return int(flag.item()) == 0

        class UniqueId(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_byte * 128)]  # rccl.h NCCL_UNIQUE_ID_BYTES

        lib, uid, ok = None, UniqueId(), True
        try:
            lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))  # the instance torch has loaded
            lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
            lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            if self.rank == 0 and lib.ncclGetUniqueId(ctypes.byref(uid)) != 0:
                ok = False
        except (OSError, AttributeError) as exc:
            log.warning("no RCCL entry points for the library's own communicator (%s)", exc)
            ok = False
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=self.device)
        self.dist.broadcast(t, src=self._global(0), group=self.group)
        if not all_agree(ok):
            log.warning("the library's own RCCL communicator could not be set up: collectives stay on torch.distributed")
            return None
        ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            ok = lib.ncclCommInitRank(ctypes.byref(handle), self.world, uid, self.rank) == 0
        # Self-test before the library relies on it: the two reductions it issues (int64 and float64 sums, the enum values
        # it passes) on known inputs, on the current stream.  A failure or a wrong answer on ANY rank leaves the hot path on
        # torch.distributed's collectives.
        if all_agree(ok):
            with torch.cuda.device(self.device):
                ti = torch.tensor([self.rank + 1, 7], dtype=torch.int64, device=self.device)
                tf = torch.tensor([0.5 * (self.rank + 1), -1.25], dtype=torch.float64, device=self.device)
                st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                rc_i = lib.ncclAllReduce(ctypes.c_void_p(ti.data_ptr()), ctypes.c_void_p(ti.data_ptr()), 2, 4, 0, handle, st)
                rc_f = lib.ncclAllReduce(ctypes.c_void_p(tf.data_ptr()), ctypes.c_void_p(tf.data_ptr()), 2, 8, 0, handle, st)
                tri = self.world * (self.world + 1) // 2
                lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_void_p]
                tg = torch.tensor([1.5 + self.rank, -2.0 * self.rank], dtype=torch.float64, device=self.device)
                tgo = torch.zeros(2 * self.world, dtype=torch.float64, device=self.device)
                rc_g = lib.ncclAllGather(ctypes.c_void_p(tg.data_ptr()), ctypes.c_void_p(tgo.data_ptr()), 2, 8, handle, st)
                want = [v for r in range(self.world) for v in (1.5 + r, -2.0 * r)]
                ok = (rc_i == 0 and rc_f == 0 and rc_g == 0 and ti.tolist() == [tri, 7 * self.world]
                      and tf.tolist() == [0.5 * tri, -1.25 * self.world] and tgo.tolist() == want)
            if all_agree(ok):
                self._rccl_lib = lib
                self._rccl = (ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p).value, handle.value,
                              ctypes.cast(lib.ncclAllGather, ctypes.c_void_p).value)
                return self._rccl
        log.warning("the library's own RCCL communicator failed its set-up or self-test: collectives stay on torch.distributed")
        if handle.value:
            lib.ncclCommDestroy(handle)
        return None

    def _global(self, r: int) -> int:
        """Group-relative rank -> global rank (send / recv / broadcast address peers by their global rank)."""
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def all_gather_f64(self, arr) -> np.ndarray:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        t = torch.as_tensor(a.reshape(-1), device=self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().reshape((self.world,) + a.shape)

    def all_reduce_max_f64(self, arr) -> np.ndarray:
        a = np.ascontiguousarray(arr, dtype=np.float64)
        t = torch.as_tensor(a.reshape(-1), device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy().reshape(a.shape)

    def all_gather_tensor(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenate equal-sized shards along dim 0 in rank order."""
        t = t.contiguous()
        if self._stage and t.is_cuda:
            return self.all_gather_tensor(t.cpu()).to(t.device)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def all_gather_into(self, out: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        if self._stage and t.is_cuda:
            out.copy_(self.all_gather_tensor(t.cpu()))
            return out
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def all_gather_ragged(self, t: torch.Tensor, counts) -> torch.Tensor:
        counts = [int(c) for c in counts]
        assert len(counts) == self.world and t.shape[0] == counts[self.rank], (counts, self.rank, tuple(t.shape))
        cap = max(counts)
        if all(c == cap for c in counts):
            return self.all_gather_tensor(t)
        pad = torch.empty((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        if t.shape[0] < cap:
            pad[t.shape[0]:] = 0
        g = self.all_gather_tensor(pad)
        return torch.cat([g[r * cap: r * cap + counts[r]] for r in range(self.world)], dim=0)

    def all_gather_i64(self, arr) -> np.ndarray:
        """Exact integers (counts, generator state words) - all_gather_f64 would round above 2^53."""
        a = np.ascontiguousarray(arr, dtype=np.int64)
        t = torch.as_tensor(a.reshape(-1), device=self.device)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().reshape((self.world,) + a.shape)

    def all_to_all_rows(self, send: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
        """Variable all-to-all along dim 0 (rows grouped by destination rank in `send`)."""
        send = send.contiguous()
        if self._stage and send.is_cuda:
            return self.all_to_all_rows(send.cpu(), send_counts, recv_counts).to(send.device)
        out = torch.empty((int(sum(recv_counts)),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        self.dist.all_to_all_single(out, send, output_split_sizes=list(map(int, recv_counts)),
                                    input_split_sizes=list(map(int, send_counts)), group=self.group)
        return out

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks, enqueued on the current stream (no host synchronisation with RCCL)."""
        if self._stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return t
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def barrier(self):
        self.dist.barrier(group=self.group)

    # A wait of UNBOUNDED length (one rank trains a flow, the others wait for its parameters) must not sit inside a collective:
    # the NCCL / gloo watchdog ends a collective that has waited its timeout (10-30 min by default).  The producer sets a key in
    # the process group's key-value store when it is done; the consumers block on that key with a day-long timeout and only then
    # enter the broadcast.  (Private accessor of torch.distributed; without it the wait degrades to the collective itself.)
    def _store(self):
        try:
            from torch.distributed import distributed_c10d as c10d

            return c10d._get_default_store()
        except Exception:
            return None

    def _signal_key(self, tag: str) -> str:
        """Key of the next signal on `tag`: namespaced by the ranks of THIS communicator's group (two sub-groups waiting on the same
        tag must not see each other's keys) and by a per-(group, tag) epoch every member advances once per signal / await pair."""
        try:
            ranks = self.dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.dist.get_world_size()))
        except Exception:
            ranks = list(range(self.world))
        ns = f"{min(ranks)}-{max(ranks)}-{len(ranks)}"
        return f"asmc/{ns}/{tag}/{self._signal_epoch(tag, bump=True)}"

    def signal(self, tag: str, error: BaseException | str | None = None) -> None:
        """Tell the waiting ranks that the long job is done - or that it FAILED (`error`): they raise instead of waiting out the
        day-long timeout.  Call it from a `finally` / `except` of the producing rank."""
        st = self._store()
        key = self._signal_key(tag)
        if st is None:
            self._warn_no_store()
            return
        st.set(key, b"ok" if error is None else ("error:" + str(error)[:400]).encode("utf-8", "replace"))

    def await_signal(self, tag: str, timeout_s: float = 86400.0) -> None:
        st = self._store()
        key = self._signal_key(tag)
        if st is None:
            self._warn_no_store()
            return
        from datetime import timedelta

        st.wait([key], timedelta(seconds=float(timeout_s)))
        val = bytes(st.get(key))
        if val.startswith(b"error:"):
            raise RuntimeError(f"rank 0 failed while the other ranks waited on '{tag}': {val[6:].decode('utf-8', 'replace')}")

    def _warn_no_store(self) -> None:
        if not self.__dict__.get("_warned_no_store"):
            self.__dict__["_warned_no_store"] = True
            import logging

            logging.getLogger(__name__).warning(
                "torch.distributed's default store is not reachable: waits of unbounded length fall back to the next collective "
                "and its watchdog timeout")

    def _signal_epoch(self, tag: str, bump: bool) -> int:
        ep = self.__dict__.setdefault("_signal_epochs", {})
        if bump:
            ep[tag] = ep.get(tag, 0) + 1
        return ep.get(tag, 0)

    # exact-cdf carry: rank r waits for the exact running sum of ranks < r, then forwards its own
    def chain_recv(self) -> float | None:
        if self.rank == 0:
            return None
        t = torch.empty(1, dtype=torch.float64, device=self.device)
        self.dist.recv(t, src=self._global(self.rank - 1), group=self.group)
        return float(t.item())

    def chain_send(self, value: float) -> None:
        if self.rank + 1 < self.world:
            t = torch.tensor([value], dtype=torch.float64, device=self.device)
            self.dist.send(t, dst=self._global(self.rank + 1), group=self.group)

    def broadcast_f64(self, value: float, src: int) -> float:
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        self.dist.broadcast(t, src=self._global(src), group=self.group)
        return float(t.item())


def default_comm(device) -> Comm:
    """TorchDistComm when a process group with world size > 1 exists, else the trivial communicator."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = torch.device(device)
            backend = dist.get_backend()
            return TorchDistComm(dev if backend == "nccl" else torch.device("cpu"))
    except Exception:
        raise
    return Comm()
