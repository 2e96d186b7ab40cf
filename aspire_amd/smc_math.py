"""Host-side scalar logic of the SMC step, written against the engine + communicator interfaces.

Everything per-particle happens in the engine (HIP kernels); this module only combines the
reduction results with the reference's scalar arithmetic, in the reference's order, so that the
beta schedule and evidence match it.  File:line citations are relative to the reference
(mj-will/aspire).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

from typing import NamedTuple

import numpy as np

BISECT_LEVELS = 4  # candidate betas per pass = 2**4 - 1 = 15 (one k-ary bisection round)


class BetaScheduleError(RuntimeError):
    """Same name/meaning as reference smc/base.py:26."""


@dataclass
class Stats:
    """Global reduction triple for one beta: m = max lw, S1 = sum e^(lw-m), S2 = sum e^(2(lw-m))."""

    m: float
    S1: float
    S2: float
    n: int  # global particle count


def global_stats(engine, comm, ll, lp, lq, beta0: float, betas, n_global: int, shifts=None) -> list[Stats]:
    """(m, S1, S2) for each candidate beta over the GLOBAL population.

    Raises the reference's ValueError when a log-weight is NaN (samples.py:1246-1247).
    Sharded: all-reduce(max) first, so every rank exponentiates against the same global maximum
    (the reference's exp(x - x.max()), utils.py:253-255), then a rank-ordered sum of the partials.
    `shifts`: stabilising shifts to use instead of searching for the maxima (one per beta; the bisection rounds
    pass the closed form m(1) (beta - beta0) / (1 - beta0), which equals the maximum up to rounding because every
    log-weight is linear in beta): saves the max pass and, sharded, its all-reduce.  No NaN census then.
    """
    betas = np.asarray(betas, dtype=np.float64)
    if shifts is not None:
        m = np.asarray(shifts, dtype=np.float64)
        sums = engine.weights_sums(ll, lp, lq, beta0, betas, m)
        if comm.sharded:
            allsums = comm.all_gather_f64(sums)
            sums = allsums[0].copy()
            for r in range(1, comm.world):
                sums = sums + allsums[r]
        return [Stats(float(m[k]), float(sums[k, 0]), float(sums[k, 1]), n_global) for k in range(betas.size)]
    if not comm.sharded:
        st = engine.weights_stats(ll, lp, lq, beta0, betas)
        n_nan = int(st[:, 3].max())
        m, S1, S2 = st[:, 0], st[:, 1], st[:, 2]
    else:
        m_loc, n_nan_loc = engine.weights_max(ll, lp, lq, beta0, betas)
        packed = comm.all_reduce_max_f64(np.concatenate([m_loc, [float(n_nan_loc)]]))
        m, n_nan = packed[:-1], int(packed[-1])
        sums = engine.weights_sums(ll, lp, lq, beta0, betas, m)
        allsums = comm.all_gather_f64(sums)  # [world, K, 2]
        tot = allsums[0].copy()
        for r in range(1, comm.world):  # rank order => identical on every rank
            tot = tot + allsums[r]
        S1, S2 = tot[:, 0], tot[:, 1]
    if n_nan > 0:
        raise ValueError(f"Log weights contain NaN values for beta={betas[0] if betas.size == 1 else betas}")
    return [Stats(float(m[k]), float(S1[k]), float(S2[k]), n_global) for k in range(betas.size)]


def log_evidence_ratio(st: Stats) -> float:
    """samples.py:1226-1228: logsumexp(log_w) - log(N); logsumexp = c + log(sum(exp(x - c)))."""
    return float((st.m + np.log(st.S1)) - math.log(st.n))


def ess(st: Stats) -> float:
    """utils.py:510-512 applied to samples.py:1244-1249's shifted log-weights.

    log_weights = lw + (LSE(lw) - log N) =: lw + c, so max = m + c and the two log-sum-exps are
    (m + c) + log S1 and 2 (m + c) + log S2; ESS = exp(2 LSE(lw') - LSE(2 lw')).
    """
    with np.errstate(all="ignore"):
        c = (st.m + np.log(st.S1)) - math.log(st.n)
        mp = st.m + c
        l1 = mp + np.log(st.S1)
        l2 = mp * 2.0 + np.log(st.S2)
        return float(np.exp(l1 * 2.0 - l2))


def evidence_variance(engine, comm, ll, lp, lq, beta0: float, beta: float, st: Stats) -> float:
    """samples.py:1230-1242: u = exp(lw - max); var(u) / (N mean(u)^2), population variance."""
    mean_u = st.S1 / st.n
    m2 = engine.weights_m2(ll, lp, lq, beta0, beta, st.m, mean_u)
    if comm.sharded:
        parts = comm.all_gather_f64(np.array([m2]))
        m2 = float(parts[0, 0])
        for r in range(1, comm.world):
            m2 = m2 + float(parts[r, 0])
    var_u = m2 / st.n
    if mean_u != 0:
        return float(var_u / (st.n * (mean_u**2)))
    return float("nan")


def evidence_variance_and_lse(engine, comm, ll, lp, lq, beta0: float, beta: float, st: Stats):
    """`evidence_variance` and, from the same pass over the batch, the sum S1' of the second log-sum-exp that
    the resampling step takes over the SHIFTED log-weights (samples.py:1277 on :1244-1249).
    Returns (variance, S1')."""
    mean_u = st.S1 / st.n
    shift = float((st.m + np.log(st.S1)) - math.log(st.n))
    mp = st.m + shift
    m2, s1p = engine.weights_m2_lse(ll, lp, lq, beta0, beta, st.m, mean_u, shift, mp)
    if comm.sharded:
        parts = comm.all_gather_f64(np.array([m2, s1p]))
        m2, s1p = float(parts[0, 0]), float(parts[0, 1])
        for r in range(1, comm.world):
            m2, s1p = m2 + float(parts[r, 0]), s1p + float(parts[r, 1])
    var_u = m2 / st.n
    var = float(var_u / (st.n * (mean_u**2))) if mean_u != 0 else float("nan")
    return var, s1p


def current_target_efficiency(target, rate: float, beta: float) -> float:
    """smc/base.py:114-121."""
    if isinstance(target, tuple):
        return target[0] + (target[1] - target[0]) * (beta**rate)
    return target


def validate_target_efficiency(value):
    """smc/base.py:84-112 (same messages)."""
    if isinstance(value, float):
        if not (0 < value < 1):
            raise ValueError("target_efficiency must be in (0, 1)")
        return value
    if len(value) != 2:
        raise ValueError("target_efficiency must be a float or tuple of two floats")
    value = tuple(map(float, value))
    if not (0 < value[0] < value[1] < 1):
        raise ValueError("target_efficiency tuple must be in (0, 1) and increasing")
    return value


def _bisection_tree(lo: float, hi: float, levels: int):
    """Heap-ordered midpoints of the next `levels` bisection steps from (lo, hi).

    node i: interval (lo_i, hi_i), mid_i = 0.5*(hi_i + lo_i) (the reference's expression,
    smc/base.py:178); child 2i+1 is taken when eff < target (hi <- mid), child 2i+2 when
    eff >= target (lo <- mid).  The values are exactly those the sequential loop would visit.
    """
    n_nodes = (1 << levels) - 1
    los, his, mids = [0.0] * n_nodes, [0.0] * n_nodes, [0.0] * n_nodes
    los[0], his[0] = lo, hi
    for i in range(n_nodes):
        mids[i] = 0.5 * (his[i] + los[i])
        l, r = 2 * i + 1, 2 * i + 2
        if r < n_nodes:
            los[l], his[l] = los[i], mids[i]
            los[r], his[r] = mids[i], his[i]
    return mids


class ScheduleRules(NamedTuple):
    beta_step: float        # fixed ladder: the increment (NaN when the schedule is adaptive)
    min_beta_step: float    # floor of an adaptive step
    max_beta_step: float    # its ceiling
    adaptive_floor: bool    # the floor is rescaled by the remaining distance to beta = 1 after every step


def schedule_rules(n_steps, adaptive, min_beta_step, max_beta_step, max_n_steps) -> ScheduleRules:
    """The rules `determine_beta` applies, from `sample()`'s keywords - the reference's set-up (smc/base.py:344-367) with
    its error behaviour: a fixed number of steps wins over `adaptive`; neither is an error; without an explicit floor,
    `max_n_steps` implies the floor 1/max_n_steps, and that one adapts (smc/base.py:198-201); the ceiling must lie in (0, 1)."""
    if n_steps is None and not adaptive:
        raise ValueError("Either n_steps or adaptive=True must be set")
    ladder = float("nan") if n_steps is None else 1 / n_steps
    floor_follows = min_beta_step is None and max_n_steps is not None
    floor = (1 / max_n_steps if floor_follows else 0.0) if min_beta_step is None else min_beta_step
    if max_beta_step is not None and not 0 < max_beta_step < 1:
        raise ValueError("max_beta_step must be in (0, 1)")
    return ScheduleRules(ladder, floor, 1.0 if max_beta_step is None else max_beta_step, floor_follows)


def determine_beta(eff_fn, beta: float, *, adaptive: bool, beta_step: float, min_beta_step: float,
                   max_beta_step: float, beta_tolerance: float, adaptive_min_beta_step: bool, target,
                   rate: float, levels: int = BISECT_LEVELS, logger=None, search_fn=None):
    """smc/base.py:123-213.  `eff_fn(betas) -> list[float]` returns ESS/N for candidate betas
    (one device pass per call).  `search_fn(beta_prev, target_eff, tol) -> (beta_star, n_passes)`, when
    given, runs the whole ESS(1.0)-check + bisection on the device (asmc_find_beta) instead.
    Returns (beta, min_beta_step, n_passes)."""
    n_pass = 0
    if not adaptive:
        beta += beta_step
        if beta >= 1.0:
            beta = 1.0
        return beta, min_beta_step, n_pass
    beta_prev = beta
    beta_min = beta_prev
    beta_max = 1.0
    current_eff = current_target_efficiency(target, rate, beta_prev)
    target_eff = current_eff
    if search_fn is not None:
        beta_min, n_pass = search_fn(beta_prev, target_eff, beta_tolerance)[:2]
        beta_max = beta_min  # converged on device
    else:
        eff_beta_max = eff_fn([beta_max])[0]
        n_pass += 1
        if eff_beta_max >= current_eff:
            beta_min = 1.0
    while beta_max - beta_min > beta_tolerance:
        mids = _bisection_tree(beta_min, beta_max, levels)
        # closed_form=True lets eff_fn shift the log-sum-exps by m(1) (beta - beta0) / (1 - beta0) instead of
        # searching for each candidate's maximum (the pass at beta = 1 above has found m(1))
        try:
            effs = eff_fn(mids, closed_form=True)
        except TypeError:
            effs = eff_fn(mids)
        n_pass += 1
        i = 0
        for _ in range(levels):
            if not (beta_max - beta_min > beta_tolerance):
                break
            beta_try = mids[i]  # == 0.5 * (beta_max + beta_min)
            if effs[i] >= target_eff:
                beta_min = beta_try
                i = 2 * i + 2
            else:
                beta_max = beta_try
                i = 2 * i + 1
    beta_star = beta_min
    if beta_star <= beta_prev + beta_tolerance and beta_prev < 1.0 and logger is not None:
        logger.warning(
            "Adaptive beta search could not find a beta above %.6g that satisfies the target "
            "efficiency %.3f within tolerance %.1e; beta may remain unchanged. Consider decreasing "
            "beta_tolerance or target_efficiency.", beta_prev, target_eff, beta_tolerance)
    if adaptive_min_beta_step and beta_star < 1.0:
        # smc/base.py:198-201.  The reference divides by (1 - beta_star) unconditionally and raises
        # ZeroDivisionError when the search jumps straight to beta* = 1 with max_n_steps set; the step size is
        # irrelevant then (beta = 1 ends the schedule), so it is simply left unchanged here.
        min_beta_step = min_beta_step * (1 - beta_prev) / (1 - beta_star)
    beta = max(beta_star, beta_prev + min_beta_step)
    beta = min(beta, beta_prev + max_beta_step, 1.0)
    if beta == beta_prev:
        raise BetaScheduleError(
            f"Beta did not increase from previous value {beta:.6g}. "
            "Adaptive beta search may have failed to find a suitable beta. "
            f"Consider adjusting beta_tolerance ({beta_tolerance}), "
            f"min_beta_step ({min_beta_step}) or "
            f"target_efficiency ({target_eff}) "
            "(values may be adaptive).")
    return beta, min_beta_step, n_pass


def pcg64_state(rng):
    """{state_hi, state_lo, inc_hi, inc_lo} of a numpy Generator if it is PCG64-backed, else None."""
    bg = getattr(rng, "bit_generator", None)
    st = getattr(bg, "state", None)
    if not isinstance(st, dict) or st.get("bit_generator") != "PCG64":
        return None
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m], dtype=np.uint64)


def draw_uniforms(engine, rng, n_total: int, j0: int, n_local: int, method: str = "multinomial"):
    """The uniforms Generator.choice would draw (samples.py:1278 -> `random(n)`), slots [j0, j0+n_local).

    PCG64 generators are continued ON DEVICE through LCG jump-ahead and the host generator is then
    advanced by the same number of draws, so its state afterwards equals the reference's.  Any other
    generator draws on the host and uploads.  systematic / stratified are opt-in extras with no
    reference counterpart."""
    if method == "multinomial":
        st = pcg64_state(rng)
        if st is not None and hasattr(engine, "uniforms_pcg64"):
            u = engine.uniforms_pcg64(st, j0, n_local) if n_local > 0 else engine.empty(0)
            rng.bit_generator.advance(n_total)
            return u
        u_all = rng.random(n_total)
        return engine.asarray(u_all[j0:j0 + n_local])
    if method == "systematic":
        u0 = float(rng.random())
        return engine.systematic_uniforms(n_local, j0, n_total, u0, None)
    if method == "stratified":
        v = rng.random(n_total)
        return engine.systematic_uniforms(n_local, j0, n_total, 0.0, engine.asarray(v[j0:j0 + n_local]))
    raise ValueError(f"Unknown resample_method: {method}")


def resample_indices(engine, comm, ll, lp, lq, beta0: float, beta: float, n_out: int, rng, *, mode: str = "exact",
                     method: str = "multinomial", uniform_weights: bool = False, st: Stats | None = None,
                     s1p: float | None = None):
    """Global ancestor indices for this rank's output slots (samples.py:1276-1278).

    w = exp(log_w - logsumexp(log_w)) with log_w = log_weights(beta);
    idx = rng.choice(N, size=n_out, replace=True, p=w)
        = searchsorted(cumsum(w)/cumsum(w)[-1], random(n_out), side="right").
    `st` / `s1p`: the (m, S1, S2) triple at `beta` and the second log-sum-exp's sum when the caller already has
    them (the sampler loop does, from the beta search and the evidence-variance pass).
    Returns (idx tensor [n_out_local] of GLOBAL indices, j0)."""
    n_local = ll.numel()
    n_global = n_local * comm.world
    if uniform_weights:
        # samples.py:1273-1274: log_w = zeros -> w = exp(0 - logsumexp(zeros)) = exp(0 - log N)
        lse = float(0.0 + np.log(np.float64(n_global)))
        w = engine.full(n_local, float(np.exp(0.0 - lse)))
    else:
        if st is None:
            st = global_stats(engine, comm, ll, lp, lq, beta0, [beta], n_global)[0]
        shift = float((st.m + np.log(st.S1)) - math.log(n_global))  # log_weights adds the log-ratio
        mp = st.m + shift
        # second log-sum-exp, over the shifted log-weights (samples.py:1277)
        if s1p is not None:
            pass
        elif not comm.sharded:
            s1p = engine.weights_sums(ll, lp, lq, beta0, [beta], [mp], [shift])[0, 0]
        else:
            parts = comm.all_gather_f64(engine.weights_sums(ll, lp, lq, beta0, [beta], [mp], [shift]))
            s1p = parts[0][0, 0]
            for r in range(1, comm.world):
                s1p = s1p + parts[r][0, 0]
        lse = float(mp + np.log(s1p))
        w = engine.normalized_weights(ll, lp, lq, beta0, beta, shift, lse)
    # cumulative sum in GLOBAL particle order.  Sharded: every rank needs the whole normalised cdf for its
    # searches anyway, so the weights (8 B/particle) are all-gathered and each rank runs the same exact
    # (sequential-order) scan over all N — no rank-to-rank dependency chain, and bitwise identical everywhere.
    if comm.sharded:
        w = comm.all_gather_tensor(w)
        if hasattr(engine, "ensure_capacity"):
            engine.ensure_capacity(w.numel(), 1)
    if hasattr(engine, "cdf_normalize_last"):  # divisor stays on the device and the division rides on the scan
        cdf, _ = engine.cdf(w, mode, 0.0, want_total=False, normalize=True)
    else:
        cdf, last = engine.cdf(w, mode, 0.0)
        engine.cdf_normalize(cdf, last)
    # this rank's output slots
    per = -(-n_out // comm.world)
    j0 = min(comm.rank * per, n_out)
    j1 = min(j0 + per, n_out)
    u = draw_uniforms(engine, rng, n_out, j0, j1 - j0, method)
    idx = engine.search(cdf, u)
    return idx, j0


# ---- sharded runs: device-side search and owner-layout resampling ------------------------------------------------
SHARD_IMBALANCE = 0.25  # owner layout is used while every rank's weight share stays within +-25 % of 1/world


def find_beta_sharded(engine, comm, ll, lp, lq, beta0: float, target_eff: float, tol: float, n_global: int):
    """`engine.find_beta` over a sharded population: every k-ary round is reduce (local kernel) -> all-gather of the
    40-double rank records -> decide (same kernel on every rank), all enqueued on the stream; the host only reads the
    result after the estimated number of rounds (include/asmc.h asmc_find_beta_shard_*).  Same return tuple."""
    import torch

    # plain rounds narrow the bracket 16x, a prediction window that holds the root by much more (csrc/asmc_bisect.h): three
    # rounds are enqueued (fewer when plain rounds reach the tolerance sooner), further ones one at a time
    rounds = min(3, max(1, int(math.ceil(math.log2((1.0 - beta0) / tol) / BISECT_LEVELS - 1e-9))))
    # record buffers live on the engine: the rounds are host-bound (three enqueues each), so no allocation per round
    bufs = engine.__dict__.setdefault("_bis_bufs", {})
    if bufs.get("world") != comm.world:
        bufs.update(world=comm.world, rec=engine.empty(40), recs=engine.empty(40 * comm.world))
    rec, recs = bufs["rec"], bufs["recs"]
    launched = 0
    out = None
    for _ in range(32):
        if hasattr(engine, "find_beta_shard_rounds"):
            engine.find_beta_shard_rounds(comm, ll, lp, lq, beta0, target_eff, tol, n_global, rec, recs, launched, rounds)
            launched = max(launched, rounds)
        while launched < rounds:
            engine.find_beta_shard_reduce(ll, lp, lq, beta0, launched, rec)
            comm.all_gather_into(recs, rec)
            engine.find_beta_shard_decide(recs, comm.world, n_global, beta0, target_eff, tol, launched)
            launched += 1
        out = engine.find_beta_shard_result()
        if out[2]:
            break
        rounds += 1
    assert isinstance(rec, torch.Tensor)
    return out


def _normalized_cdf(engine, w, mode: str):
    """cumsum(w) / cumsum(w)[-1] of one array (numpy's cdf inside Generator.choice)."""
    if hasattr(engine, "cdf_normalize_last"):  # divisor stays on the device, the division rides on the scan
        return engine.cdf(w, mode, 0.0, want_total=False, normalize=True)[0]
    cdf, last = engine.cdf(w, mode, 0.0)
    engine.cdf_normalize(cdf, last)
    return cdf


def global_cdf_slice_replicated(engine, comm, w, counts, mode: str = "exact"):
    """This rank's slice of the normalised GLOBAL cdf by brute force: all-gather of the weights (8 B/particle), the same
    scan over all N on every rank, slice.  Returns (cdf_slice, edges) with edges = device tensor {fail = 0, 0, lo, hi}.
    Used by `fast` mode, by engines without tile records (the CPU test double) and as the fallback of
    `global_cdf_slice` when a tile record fails verification."""
    import torch

    w_all = comm.all_gather_ragged(w, counts)
    if hasattr(engine, "ensure_capacity"):
        engine.ensure_capacity(w_all.numel(), 1)
    cdf_all = _normalized_cdf(engine, w_all, mode)
    a = int(sum(counts[:comm.rank]))
    b = a + int(counts[comm.rank])
    zero = torch.zeros(1, dtype=cdf_all.dtype, device=cdf_all.device)
    edges = torch.cat([zero, zero, cdf_all[a - 1:a] if a > 0 else zero, cdf_all[b - 1:b]])
    return cdf_all[a:b], edges


def _global_cdf_chain(engine, comm, w, counts, approx_carry, tile_sums=None):
    """The enqueue-only part of `global_cdf_slice` up to the second chain round: (cdf buffer, all ranks' tile records, this
    rank's first global tile, chain scratch, final chain state)."""
    if tile_sums is not None:
        cdf, rec = engine.cdf_shard_records(w, approx_carry, comm.rank == 0, tile_sums)
    else:
        cdf, rec = engine.cdf_shard_records(w, approx_carry, comm.rank == 0)
    tiles = [-(-int(c) // 2048) for c in counts]
    recs_all = comm.all_gather_ragged(rec, tiles).contiguous()
    tile0 = int(sum(tiles[:comm.rank]))
    # round 1: every rank walks the chain; the owner of a tile that fails verification (the tail of normalised weights,
    # typically) scans it element-wise and publishes the exact sum behind it, the others stop in front of it
    work, state = engine.cdf_shard_chain(w, cdf, recs_all, tile0, None, None, comm.world, comm.rank)
    states = (engine.all_gather(comm, state) if hasattr(engine, "all_gather") else comm.all_gather_tensor(state)).contiguous()
    # round 2: resume through the published sums
    work, state = engine.cdf_shard_chain(w, cdf, recs_all, tile0, work, states, comm.world, comm.rank)
    return cdf, recs_all, tile0, work, state


def global_cdf_slice(engine, comm, w, counts, approx_carry: float, mode: str = "exact", force_replicated: bool = False):
    """This rank's slice of numpy's sequential cumsum over the GLOBAL weight vector, divided by the global total
    (samples.py:1277-1278 -> Generator.choice), bit for bit, plus the slice's edges {fail, total, lo, hi} on the device.
    Exact mode: every rank turns its shard into per-tile records from an approximate incoming sum, the records are
    all-gathered (72 B per 2048 particles) and every rank walks the same verifying chain over all of them, in two
    rounds with one all-gather of the ranks' chain states in between (include/asmc.h asmc_cdf_shard_*): no rank waits
    for another rank's scan.  edges[0] != 0: the chain could not be closed - the caller repeats with the replicated scan."""
    if force_replicated or mode != "exact" or not hasattr(engine, "cdf_shard_records"):
        return global_cdf_slice_replicated(engine, comm, w, counts, mode)
    cdf, recs_all, tile0, work, state = _global_cdf_chain(engine, comm, w, counts, approx_carry)
    edges = engine.cdf_shard_finish(w, cdf, recs_all, tile0, work, state)
    return cdf, edges


def resample_owner(engine, comm, ll, lp, lq, beta0: float, beta: float, n_out: int, rng, *, mode: str = "exact",
                   st: Stats | None, counts=None, method: str = "multinomial", uniform_weights: bool = False,
                   force_replicated: bool = False):
    """Sharded resampling that keeps every offspring on its ancestor's rank and still selects EXACTLY the reference's
    ancestors (samples.py:1276-1278: `rng.choice(N, n_out, p=w)` = searchsorted(cumsum(w) / cumsum(w)[-1], random(n_out))).

    Every rank holds its slice of the global sequential cdf (`global_cdf_slice`), walks the same n_out draws of the
    generator (the ranks' generators must be in the same state: `sync_rng`), keeps the draws u with
    C_r <= u < C_{r+1} - exactly those whose ancestor it owns - in DRAW ORDER, and searches its slice: rank r ends up with
    the sub-sequence of Generator.choice's index vector that points into its shard.  No particle row crosses a link;
    shards become ragged (`counts`, the rows per rank, travel with the population).  Collectives: one all-gather of two
    doubles (second log-sum-exp + evidence-variance partials, samples.py:1230-1242), one of the tile records, one of the
    kept counts.
    Returns (idx_local or None, variance, s1p, new_counts): idx_local is None when a rank's weight share is outside
    1/world (1 +- SHARD_IMBALANCE) - the caller then uses the slot layout (`resample_indices` + row exchange), which
    rebalances the shards; every rank takes that decision from the same gathered numbers."""
    world, rank, n_local = comm.world, comm.rank, ll.numel()
    counts = [n_local] * world if counts is None else [int(c) for c in counts]
    assert counts[rank] == n_local, (counts, rank, n_local)
    var, s1p = None, None
    if uniform_weights:
        # samples.py:1273-1274: log_w = zeros -> w = exp(0 - logsumexp(zeros)) = exp(0 - log N)
        n_global = int(sum(counts))
        lse = float(0.0 + np.log(np.float64(n_global)))
        w = engine.full(n_local, float(np.exp(0.0 - lse)))
        approx_carry = float(sum(counts[:rank])) / n_global
    else:
        mean_u = st.S1 / st.n
        shift = float((st.m + np.log(st.S1)) - math.log(st.n))
        mp = st.m + shift
        if hasattr(engine, "weights_m2_lse_dev"):
            rec = engine.empty(2)
            engine.weights_m2_lse_dev(ll, lp, lq, beta0, beta, st.m, mean_u, shift, mp, rec)
            parts = engine.to_numpy(engine.all_gather(comm, rec) if hasattr(engine, "all_gather")
                                    else comm.all_gather_tensor(rec)).reshape(world, 2)
        else:
            parts = comm.all_gather_f64(np.array(engine.weights_m2_lse(ll, lp, lq, beta0, beta, st.m, mean_u, shift, mp)))
        m2, s1p = float(parts[0, 0]), float(parts[0, 1])
        for r in range(1, world):  # rank order: the same floats on every rank
            m2, s1p = m2 + float(parts[r, 0]), s1p + float(parts[r, 1])
        var_u = m2 / st.n
        var = float(var_u / (st.n * (mean_u**2))) if mean_u != 0 else float("nan")
        share = parts[:, 1] / s1p
        if not (np.all(share * world <= 1.0 + SHARD_IMBALANCE) and np.all(share * world >= 1.0 - SHARD_IMBALANCE)):
            return None, var, s1p, None
        lse = float(mp + np.log(s1p))  # second log-sum-exp, over the shifted log-weights (samples.py:1277)
        w = engine.normalized_weights(ll, lp, lq, beta0, beta, shift, lse)
        approx_carry = float(np.sum(parts[:rank, 1]) / s1p) if rank > 0 else 0.0
    # every rank draws ALL n_out uniforms (PCG64: regenerated on the device by jump-ahead, 8 B per draw)
    u_all = draw_uniforms(engine, rng, int(n_out), 0, int(n_out), method)
    for attempt in range(2):
        cdf, edges = global_cdf_slice(engine, comm, w, counts, approx_carry, mode, force_replicated or attempt == 1)
        if hasattr(engine, "select_range_dev") and edges.is_contiguous() and edges.numel() == 4:
            # count and failure flag stay on the device until the ranks' pairs have been gathered: one synchronisation
            buf, info_dev = engine.select_range_dev(u_all, edges)
            info = engine.to_numpy(engine.all_gather(comm, info_dev) if hasattr(engine, "all_gather")
                                   else comm.all_gather_tensor(info_dev)).reshape(world, 2)
            u_kept = buf[: int(info[rank, 0])]
        else:
            u_kept = engine.select_range(u_all, edges[2:4])  # synchronises: everything above is enqueued by now
            info = comm.all_gather_i64(np.array([u_kept.numel(), int(round(float(edges[0].item())))], dtype=np.int64))
        if not info[:, 1].any():
            break
        if attempt == 1:
            raise RuntimeError("sharded cdf: the replicated scan reported a failure")
    new_counts = [int(c) for c in info[:, 0]]
    if sum(new_counts) != int(n_out):
        raise RuntimeError(f"owner-layout resampling kept {sum(new_counts)} of {n_out} draws: the ranks' generators are not "
                           "in the same state (pass the same seeded rng to every rank; HipSMC.sample synchronises it)")
    if min(new_counts) == 0:  # decided from the gathered counts: every rank raises, nobody is left waiting in a collective
        raise RuntimeError(f"owner-layout resampling left rank {new_counts.index(0)} without offspring")
    return engine.search(cdf, u_kept), var, s1p, new_counts


def shard_step_available(engine, comm) -> bool:
    """The sharded importance step can run as one chain of launches (`shard_step_enqueue`): the engine has the passes that
    take their scalars from the device, and ASMC_SHARD_STEP=0 has not switched it off (A-B switch / escape hatch)."""
    return (comm.sharded and all(hasattr(engine, k) for k in ("weights_m2_lse_shard", "normalized_weights_shard", "all_gather",
                                                              "shard_step_result", "find_beta_shard_round", "select_range_dev"))
            and os.environ.get("ASMC_SHARD_STEP", "1") != "0")


def shard_step_enqueue(engine, comm, ll, lp, lq, beta0: float, target_eff: float, tol: float, n_global: int, counts,
                       state4, n_out: int):
    """The importance step of a SHARDED population up to the ranks' offspring counts - adaptive-beta search
    (smc/base.py:167-186), evidence-variance and second log-sum-exp partials (samples.py:1230-1242, :1277), normalised weights,
    this rank's slice of the global sequential cdf, the draws of Generator.choice that fall into it (samples.py:1278) - as ONE
    chain of launches and collectives with no host decision in between: what `find_beta_sharded` and `resample_owner` read
    back between their phases (beta*, the weight sums, the shares) stays on the device and parameterises the next launch
    (include/asmc.h asmc_weights_m2_lse_shard).  Nothing here synchronises; `shard_step_wait` does, once.
    `state4`: the PCG64 state words of the generator every rank holds in the same state (`sync_rng`).  Returns the handle for
    `shard_step_wait` / `shard_step_check`."""
    world, rank, n_local = comm.world, comm.rank, ll.numel()
    counts = [int(c) for c in counts]
    assert counts[rank] == n_local and sum(counts) == int(n_global), (counts, rank, n_local, n_global)
    import torch

    # search: the rounds `find_beta_sharded` enqueues before it looks (plain rounds narrow the bracket 16x each)
    # (an engine whose rounds are all plain - no prediction windows: the CPU test double - says how many it may need)
    rounds = min(getattr(engine, "search_rounds_cap", 3),
                 max(1, int(math.ceil(math.log2((1.0 - beta0) / tol) / BISECT_LEVELS - 1e-9))))
    bufs = engine.__dict__.setdefault("_bis_bufs", {})
    if bufs.get("world") != world:
        bufs.update(world=world, rec=engine.empty(40), recs=engine.empty(40 * world))
    rec, recs = bufs["rec"], bufs["recs"]
    for r in range(rounds):  # one launch + one all-gather per round: round r closes round r - 1 itself (no decide launch)
        engine.find_beta_shard_round(ll, lp, lq, beta0, target_eff, tol, world, n_global, r, recs if r else None, rec)
        engine.all_gather(comm, rec, out=recs)
    # evidence moments + second log-sum-exp at the beta the last decide step left on the device; the gathered pairs, the
    # search state and (below) the ranks' offspring counts share ONE buffer: the step's single read-back
    res = engine.empty(40 + 4 * world)
    part = engine.empty(2)
    engine.weights_m2_lse_shard(ll, lp, lq, part, recs, world, n_global, beta0, target_eff, tol, rounds)  # (closes the last round)
    parts = engine.all_gather(comm, part, out=res[40:40 + 2 * world])
    w, carry, tile_sums = engine.normalized_weights_shard(ll, lp, lq, parts, world, rank, float(sum(counts[:rank])) / float(n_global),
                                                          state_copy=res[:40])
    rec_token = getattr(engine, "rec_token", 0)  # (the gather's records were packed on the way: engine.rec_claim)
    u_all = engine.uniforms_pcg64(state4, 0, int(n_out))  # every rank walks ALL n_out draws
    if hasattr(engine, "cdf_shard_finish_select"):  # write pass + edges / count + scan / scatter / info: three launches
        cdf, recs_all, tile0, work, state = _global_cdf_chain(engine, comm, w, counts, carry, tile_sums)
        edges, buf, info_dev = engine.cdf_shard_finish_select(w, cdf, recs_all, tile0, work, state, u_all)
        keep = (w, u_all, edges, recs_all, work, state, tile_sums, carry, part, info_dev)
    else:  # engines without the tile-record machinery (the CPU test double): the same slice through the replicated scan
        cdf, edges = global_cdf_slice(engine, comm, w, counts, float(carry[0]), "exact")
        buf, info_dev = engine.select_range_dev(u_all, edges)
        keep = (w, u_all, edges, carry, part, info_dev)
    engine.all_gather(comm, info_dev, out=res[40 + 2 * world:].view(torch.int64))
    return dict(res=res, cdf=cdf, buf=buf, rounds=rounds, world=world, rank=rank, n_out=int(n_out), n_global=int(n_global),
                rec_token=rec_token, keep=keep)


def shard_step_wait(engine, comm, h):
    """Wait for `shard_step_enqueue`'s chain (its ONE synchronisation).  Returns (search tuple, parts[world, 2], info[world, 2],
    kept): `kept` = this rank's draws (a view of the selection buffer) when the numbers read back are those of a finished step
    - the caller may enqueue the search and the gather behind them right away, before `shard_step_check` has looked at the rest
    (the GPU then does not idle while the host decides) - else None."""
    search, parts, info = engine.shard_step_result(h["res"], h["world"])
    _, _, converged, _, n_nan, trip, _ = search
    cnt = int(info[h["rank"], 0])
    # (a rank-UNIFORM condition - every rank reads the same gathered numbers: what follows contains collectives)
    usable = (converged and n_nan == 0 and trip is not None and not info[:, 1].any()
              and int(info[:, 0].min()) > 0 and int(info[:, 0].max()) <= h["buf"].numel())
    return search, parts, info, (h["buf"][:cnt] if usable else None)


def shard_step_check(h, search, parts, info):
    """The decisions `find_beta_sharded` / `resample_owner` take, from the same numbers in the same order (every rank reads the
    same gathered values, so every rank decides alike).  Returns (ok, m2, s1p, new_counts): `ok` False - search not converged in
    the rounds enqueued, NaN weights, a weight share outside 1/world (1 +- SHARD_IMBALANCE), a cdf chain that needs the
    replicated scan, a rank left without offspring - means the caller runs the step phase by phase (which handles or reports
    each of these); the generator has not been touched."""
    world = h["world"]
    _, _, converged, _, n_nan, trip, _ = search
    if not converged or n_nan > 0 or trip is None:
        return False, None, None, None
    m2, s1p = float(parts[0, 0]), float(parts[0, 1])
    for r in range(1, world):  # rank order: the same floats on every rank
        m2, s1p = m2 + float(parts[r, 0]), s1p + float(parts[r, 1])
    share = parts[:, 1] / s1p if s1p > 0 else np.full(world, np.nan)
    ok = bool(np.all(share * world <= 1.0 + SHARD_IMBALANCE) and np.all(share * world >= 1.0 - SHARD_IMBALANCE))
    new_counts = [int(c) for c in info[:, 0]]
    ok = ok and not info[:, 1].any() and sum(new_counts) == h["n_out"] and min(new_counts) > 0
    return ok, m2, s1p, new_counts


def sync_rng(comm, rng):
    """Put every rank's generator into rank 0's state (sharded resampling walks the same draws on every rank; mutation
    seeds are drawn from it).  PCG64 generators get the exact state words; anything else is re-seeded from an integer
    that rank 0 draws.  Returns the generator to use."""
    if not comm.sharded:
        return rng
    st = pcg64_state(rng)
    # one exchange: {is PCG64, the four state words} of every rank
    mine = np.concatenate([[0 if st is None else 1], (st.view(np.int64) if st is not None else np.zeros(4, dtype=np.int64))]).astype(np.int64)
    allst = comm.all_gather_i64(mine)
    if allst[:, 0].all():
        words = np.ascontiguousarray(allst[0, 1:]).view(np.uint64)
        state = rng.bit_generator.state
        state["state"]["state"] = (int(words[0]) << 64) | int(words[1])
        state["state"]["inc"] = (int(words[2]) << 64) | int(words[3])
        state["has_uint32"], state["uinteger"] = 0, 0
        rng.bit_generator.state = state
        return rng
    seed = int(comm.all_gather_i64(np.array([int(rng.integers(0, 2**62))], dtype=np.int64))[0, 0])
    return np.random.default_rng(seed)


def owner_layout_ok(engine, comm, rng, method: str, uniform_weights: bool = False) -> bool:
    """Owner layout: any resampling method, any generator - every rank walks the same draws (`sync_rng`)."""
    return comm.sharded and hasattr(engine, "select_range")
