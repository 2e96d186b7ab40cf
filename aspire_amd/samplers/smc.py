"""SMC samplers: the temperature loop and the pCN mutation on MI355X.

`SMCSampler`  mirrors reference src/aspire/samplers/smc/base.py:30-562 (loop order, defaults,
              exceptions, history fields) with the particle state resident in HBM.
`HipSMC`      mirrors reference src/aspire/samplers/smc/minipcn.py:14-135 (`"smc"` / `"minipcn_smc"`):
              the mutation kernel is this repository's fused pCN HIP kernel instead of the
              third-party `minipcn` package (parity unpinned — see DESIGN.md §pCN).
"""
from __future__ import annotations

import copy
import logging
import math
import os
from typing import Any, Callable

import numpy as np
import torch

from .. import smc_math
from .._xp import is_torch, to_numpy
from ..history import SMCHistory
from ..samples import SMCSamples, Samples
from ..targets import DiagGaussianMixture
from .base import IdentityTransform, MCMCSampler, track_calls

logger = logging.getLogger(__name__)

DEFAULT_BETA_TOLERANCE = 1e-8  # smc/base.py:23


class SMCSampler(MCMCSampler):
    """smc/base.py:30-562."""

    def __init__(self, log_likelihood: Callable, log_prior: Callable, dims: int, prior_flow, xp: Callable,
                 dtype: Any | str | None = None, parameters: list[str] | None = None, rng=None,
                 preconditioning_transform: Callable | None = None, engine=None, comm=None):
        super().__init__(log_likelihood=log_likelihood, log_prior=log_prior, dims=dims, prior_flow=prior_flow,
                         xp=xp, dtype=dtype, parameters=parameters,
                         preconditioning_transform=preconditioning_transform, rng=rng, engine=engine, comm=comm)
        self._adaptive_target_efficiency = False
        self.device_bisection = True  # single-rank: whole adaptive-beta search on device (asmc_find_beta)
        self.fused_importance_step = True  # single-rank: search + moments + resampling enqueued as one chain
        self.last_mutation_path = None  # which of the mutation code paths the last `mutate` call took (a description; `engine.profile` names the kernels)
        self.resample_mode = "exact"
        self.resample_method = "multinomial"
        self.shard_layout = "owner"  # sharded runs: offspring stay on the ancestor's rank ("slots": single-rank order)

    # ---- target efficiency (smc/base.py:80-121) -------------------------------------------
    @property
    def target_efficiency(self):
        return self._target_efficiency

    @target_efficiency.setter
    def target_efficiency(self, value):
        value = smc_math.validate_target_efficiency(value)
        self._target_efficiency = value
        self._adaptive_target_efficiency = isinstance(value, tuple)

    def current_target_efficiency(self, beta: float) -> float:
        return smc_math.current_target_efficiency(self._target_efficiency, self.target_efficiency_rate, beta)

    # ---- helpers --------------------------------------------------------------------------
    def _n_global(self, samples) -> int:
        return samples._n_global()

    def _gid0(self, samples) -> int:
        """Global index of this rank's first particle (keys the per-particle noise streams): a sharded run in slot
        layout draws the same noise as a single-rank run; ragged shards (owner layout) count the rows of the lower
        ranks, so the id ranges stay disjoint and contiguous."""
        if hasattr(samples, "gid0"):
            return samples.gid0()
        return self.comm.rank * len(samples.x)

    def _stats(self, samples: SMCSamples, betas) -> list[smc_math.Stats]:
        return samples.weight_stats(betas)

    def _global_counts(self, local_counts) -> list[int]:
        """Element-wise sum of small integer censuses over the ranks (NaN guards must agree on every rank)."""
        a = np.asarray(local_counts, dtype=np.int64)
        if not self.comm.sharded:
            return [int(v) for v in a]
        return [int(v) for v in self.comm.all_gather_i64(a).sum(axis=0)]

    def _resample_moments_n(self, samples) -> int | None:
        """Population size when the mutation behind this resampling fits its reference Gaussian to the moments of the whole
        resampled population (see HipSMC): the resampling call then starts them behind its gather.  None otherwise."""
        return None

    def _importance_step_follows(self, beta: float) -> bool:
        """Another temperature follows this mutation and its importance step runs as the fused chain of launches (single
        rank): the mutation may then enqueue that chain behind itself."""
        if os.environ.get("ASMC_IS_AHEAD", "1") == "0":  # escape hatch / A-B switch
            return False
        return (self.fused_importance_step and getattr(self, "adaptive", False) and self.device_bisection and beta < 1.0
                and not getattr(self, "_last_iteration", False) and hasattr(self.engine, "pcn_mutate_flow_enqueue")
                and ((self.shard_layout == "owner" and smc_math.shard_step_available(self.engine, self.comm)) if self.comm.sharded
                     else (hasattr(self.engine, "importance_step") and not getattr(self.engine, "importance_step_disabled", False)))
                and getattr(self, "_beta_tolerance", None) is not None)

    def _speculated_moments_n(self, samples) -> int | None:
        """Population size for the moments the fused importance step may compute behind its gather (see HipSMC); None:
        this sampler's mutation does not use them."""
        return None

    def _factor_ahead(self) -> bool:
        """Whether the speculated importance step may also enqueue the factorisation of the reference Gaussian (see HipSMC)."""
        return False

    def determine_beta(self, samples: SMCSamples, beta: float, beta_step: float, min_beta_step: float,
                       max_beta_step: float = 1.0, beta_tolerance: float = DEFAULT_BETA_TOLERANCE):
        """smc/base.py:123-213; the ESS evaluations of one k-ary bisection round share a device pass."""
        n = self._n_global(samples)

        def eff_fn(betas, closed_form=False):
            if closed_form and 1.0 in samples.__dict__.get("_wstats", {}):
                m_one, b0 = samples.weight_stats([1.0])[0].m, float(samples.beta)
                shifts = [m_one * ((b - b0) / (1.0 - b0)) for b in betas]
                sts = smc_math.global_stats(self.engine, self.comm, samples.log_likelihood, samples.log_prior, samples.log_q,
                                            b0, betas, n, shifts=shifts)
                return [smc_math.ess(s) / n for s in sts]
            return [smc_math.ess(s) / n for s in self._stats(samples, betas)]

        search_fn = None
        sharded = self.comm.sharded and hasattr(self.engine, "find_beta_shard_reduce")
        spec0 = samples.__dict__.get("_spec")
        have_spec = spec0 is not None and spec0["search"][2] and float(beta) == float(samples.beta)
        if (((not self.comm.sharded and hasattr(self.engine, "find_beta")) or sharded or have_spec) and self.device_bisection
                and beta < 1.0):
            def search_fn(beta_prev, target_eff, tol):
                spec = samples.__dict__.get("_spec")
                if (spec is not None and spec["key"] == (float(target_eff), float(tol)) and spec["search"][2]
                        and float(beta_prev) == float(samples.beta)):
                    b, _, converged, passes, n_nan, trip, trip_one = spec["search"]  # the fused step already searched
                elif sharded:
                    b, _, converged, passes, n_nan, trip, trip_one = smc_math.find_beta_sharded(
                        self.engine, self.comm, samples.log_likelihood, samples.log_prior, samples.log_q, float(beta_prev),
                        float(target_eff), float(tol), n)
                else:
                    b, _, converged, passes, n_nan, trip, trip_one = self.engine.find_beta(
                        samples.log_likelihood, samples.log_prior, samples.log_q, float(beta_prev), float(target_eff),
                        float(tol))
                if n_nan > 0:
                    raise ValueError(f"Log weights contain NaN values for beta={b}")
                if not converged:
                    raise RuntimeError("device-side beta search did not converge")
                # the search already reduced the batch at beta* and at 1.0: keep those triples for the ESS /
                # evidence / resampling steps of this iteration
                if trip is not None:
                    samples.remember_stats(b, smc_math.Stats(*trip, n))
                samples.remember_stats(1.0, smc_math.Stats(*trip_one, n))
                return b, passes

        beta, min_beta_step, _ = smc_math.determine_beta(
            eff_fn, beta, adaptive=self.adaptive, beta_step=beta_step, min_beta_step=min_beta_step,
            max_beta_step=max_beta_step, beta_tolerance=beta_tolerance,
            adaptive_min_beta_step=self.adaptive_min_beta_step, target=self._target_efficiency,
            rate=self.target_efficiency_rate, logger=logger, search_fn=search_fn)
        return beta, min_beta_step

    def _wrap(self, x, ll, lp, lq, beta, like: SMCSamples | None = None) -> SMCSamples:
        s = SMCSamples(x=x, xp=torch, beta=beta, parameters=self.parameters, engine=self.engine, comm=self.comm)
        s.log_likelihood, s.log_prior, s.log_q = ll, lp, lq
        if like is not None:  # shard bookkeeping of the population these particles came from
            for k in ("n_global", "ragged", "shard_counts"):
                if k in like.__dict__:
                    setattr(s, k, like.__dict__[k])
        return s

    # ---- the loop (smc/base.py:215-488) -----------------------------------------------------
    @track_calls
    def sample(self, n_samples: int, n_steps: int | None = None, adaptive: bool = True,
               min_beta_step: float | None = None, max_beta_step: float | None = None,
               max_n_steps: int | None = None, target_efficiency: float = 0.5,
               target_efficiency_rate: float = 1.0, n_final_samples: int | None = None,
               checkpoint_callback: Callable[[dict], None] | None = None, checkpoint_every: int | None = None,
               checkpoint_file_path: str | None = None, resume_from: str | bytes | dict | None = None,
               store_sample_history: bool = True, beta_tolerance: float = DEFAULT_BETA_TOLERANCE,
               resample_mode: str | None = None, resample_method: str | None = None):
        if resample_mode is not None:
            self.resample_mode = resample_mode
        if resample_method is not None:
            self.resample_method = resample_method
        comm = self.comm
        if n_samples % comm.world:
            raise ValueError(f"n_samples ({n_samples}) must be divisible by the number of ranks ({comm.world})")
        n_local = n_samples // comm.world
        if hasattr(self.engine, "ensure_capacity"):
            nf = n_final_samples or 0
            self.engine.ensure_capacity(max(n_samples if comm.sharded else n_local, nf), self.dims)
        if hasattr(self.prior_flow, "gid0"):
            self.prior_flow.gid0 = comm.rank * n_local
        if hasattr(self.prior_flow, "attach_engine") and (getattr(self, "sampler_kwargs", None) or {}).get("flow_sample_on_engine", True):
            # the proposal draw itself on the engine (asmc_coupling_sample), emitted in the particles' dtype
            self.prior_flow.attach_engine(self.engine, getattr(self, "x_torch_dtype", None), comm.rank * n_local)
        # sharded runs: every rank walks the same resampling draws and draws the same mutation seeds, so the ranks'
        # generators must be in the same state - rank 0's is handed to everyone (a no-op for one rank)
        self.rng = smc_math.sync_rng(comm, self.rng)
        if comm.sharded and hasattr(self.engine, "use_rccl"):
            # the library's own communicator (comm.rccl_direct) is set up by a collective: here, on every rank, once - never
            # lazily behind a condition that one rank might evaluate differently from the others
            self.engine.use_rccl(comm)
        if comm.sharded and hasattr(self.prior_flow, "sync_shards"):
            self.prior_flow.sync_shards(comm)  # the same trained flow on every rank, a separate draw stream per rank
        if getattr(self.prior_flow, "seed_from_rng", False) and resume_from is None:
            # the proposal's own draw stream follows the run's generator (in the reference flow sampling is stochastic per
            # run); drawn after the synchronisation, so every rank gets the same seed and particle ids keep shards apart
            self.prior_flow.seed = int(self.rng.integers(0, 2**63 - 1, dtype=np.int64))
        resumed = resume_from is not None
        if resumed:
            samples, beta, iterations = self.restore_from_checkpoint(resume_from)
            logger.info(f"Resumed SMC sampling at iteration {iterations} with beta={beta:.4f}")
        else:
            init = self.draw_initial_samples(n_local)
            samples = self._wrap(init.x, init.log_likelihood, init.log_prior, init.log_q, 0.0)
            beta = 0.0
            iterations = 0
            self.history = SMCHistory()
        self.fit_preconditioning_transform(samples.x)

        if store_sample_history:
            self.history.sample_history.append(samples.to_numpy())

        e = self.engine
        # one exchange for the three censuses: every rank raises (or none does), nobody is left waiting in a collective
        nans = self._global_counts([e.count_nonfinite(arr)[0] for arr in (samples.log_q, samples.log_prior,
                                                                           samples.log_likelihood)])
        for name, n_nan in zip(("Log proposal", "Log prior", "Log likelihood"), nans):
            if n_nan:
                raise ValueError(f"{name} contains NaN values")

        self.sampler_kwargs = dict(getattr(self, "sampler_kwargs", None) or {})
        n_final_steps = self.sampler_kwargs.pop("n_final_steps", None)  # (a loop option, not a mutation-kernel option)
        self.target_efficiency, self.target_efficiency_rate = target_efficiency, target_efficiency_rate
        # the temperature schedule's rules (smc/base.py:344-367, error texts included): a fixed ladder of 1/n_steps, or the
        # ESS search between a floor (1/max_n_steps when only that is given: that floor follows the remaining distance) and a
        # ceiling
        rules = smc_math.schedule_rules(n_steps, adaptive, min_beta_step, max_beta_step, max_n_steps)
        beta_step, min_beta_step = rules.beta_step, rules.min_beta_step
        self.adaptive, self.adaptive_min_beta_step, self.max_beta_step = adaptive, rules.adaptive_floor, rules.max_beta_step
        iterations = iterations or 0
        if checkpoint_callback is None and checkpoint_every is not None:
            checkpoint_callback = self.default_file_checkpoint_callback(checkpoint_file_path)
        if checkpoint_callback is not None and checkpoint_every is None:
            checkpoint_every = 1

        run_smc_loop = True
        if resumed:
            last_beta = self.history.beta[-1] if self.history.beta else beta
            if last_beta >= 1.0:
                run_smc_loop = False

        def maybe_checkpoint(force: bool = False):
            if checkpoint_callback is None:
                return
            should = force or (checkpoint_every is not None and checkpoint_every > 0
                               and iterations % checkpoint_every == 0)
            if not should:
                return
            checkpoint_callback(self.build_checkpoint_state(samples, iterations, beta))

        if run_smc_loop:
            while True:
                iterations += 1
                self._beta_tolerance = beta_tolerance
                # the loop ends after this iteration's mutation when max_n_steps is reached: nothing may run ahead then
                self._last_iteration = max_n_steps is not None and iterations >= max_n_steps
                if hasattr(samples, "finish_speculation"):
                    samples.finish_speculation()  # an importance step the mutation enqueued behind itself
                if (self.fused_importance_step and self.adaptive and self.device_bisection and beta < 1.0
                        and "_spec" not in samples.__dict__):  # (a mutation may already have run the step behind itself)
                    # search + evidence moments + resampling at beta* in one chain of launches, one synchronisation;
                    # determine_beta and resample below pick the parked results up (or redo the step when the
                    # schedule rules choose another beta)
                    samples.speculate_importance_step(self.current_target_efficiency(beta), beta_tolerance, self.rng,
                                                      resample_mode=self.resample_mode,
                                                      resample_method=self.resample_method,
                                                      moments_n=self._speculated_moments_n(samples),
                                                      shard_layout=self.shard_layout, factor_ahead=self._factor_ahead())
                beta, min_beta_step = self.determine_beta(samples, beta, beta_step, min_beta_step,
                                                          max_beta_step=self.max_beta_step,
                                                          beta_tolerance=beta_tolerance)
                self.history.eff_target.append(float(self.current_target_efficiency(beta)))
                logger.info(f"it {iterations} - beta: {beta}")
                self.history.beta.append(float(beta))

                # ESS(beta), ESS(1.0) and the evidence ratio share one pass (K = 2)
                st_beta, st_one = self._stats(samples, [beta, 1.0])
                ess = smc_math.ess(st_beta)
                eff = ess / self._n_global(samples)
                if eff < 0.1:
                    logger.warning(f"it {iterations} - Low sample efficiency: {eff:.2f}")
                self.history.ess.append(float(ess))
                logger.info(f"it {iterations} - ESS: {ess:.1f} ({eff:.2f} efficiency)")
                self.history.ess_target.append(float(smc_math.ess(st_one)))

                log_evidence_ratio = smc_math.log_evidence_ratio(st_beta)
                # the evidence-variance pass (samples.py:1230-1242) shares its reduction with the resampling step
                samples, log_evidence_ratio_var = samples.resample(
                    beta, rng=self.rng, resample_mode=self.resample_mode, resample_method=self.resample_method,
                    shard_layout=self.shard_layout, want_variance=True, moments_n=self._resample_moments_n(samples))
                self.history.log_norm_ratio.append(float(log_evidence_ratio))
                self.history.log_norm_ratio_var.append(float(log_evidence_ratio_var))
                logger.info(f"it {iterations} - Log evidence ratio: {log_evidence_ratio:.2f} +/- "
                            f"{np.sqrt(log_evidence_ratio_var):.2f}")
                samples = self.mutate(samples, beta)
                if store_sample_history:
                    self.history.sample_history.append(samples.to_numpy())
                maybe_checkpoint()
                if beta == 1.0 or (max_n_steps is not None and iterations >= max_n_steps):
                    break

        if n_final_samples is not None and self._n_global(samples) != n_final_samples:
            logger.info(f"Generating {n_final_samples} final samples")
            bad = self._global_counts([sum(e.count_nonfinite(arr)) for arr in (samples.log_likelihood, samples.log_prior,
                                                                                  samples.log_q)])
            for name, n_bad in zip(("log likelihood", "log prior", "log proposal"), bad):
                if n_bad:
                    logger.warning(f"Final samples contain non-finite {name} values")
            final_samples = samples.resample(1.0, n_samples=n_final_samples, rng=self.rng,
                                             resample_mode=self.resample_mode,
                                             resample_method=self.resample_method, shard_layout=self.shard_layout)
            samples = self.mutate(final_samples, 1.0, n_steps=n_final_steps)

        samples.log_evidence = float(np.sum(np.asarray(self.history.log_norm_ratio, dtype=np.float64)))
        samples.log_evidence_error = float(np.sqrt(np.sum(np.asarray(self.history.log_norm_ratio_var,
                                                                     dtype=np.float64))))
        maybe_checkpoint(force=True)

        final_samples = samples.to_standard_samples()
        logger.info(f"Log evidence: {final_samples.log_evidence:.2f} +/- {final_samples.log_evidence_error:.2f}")
        return final_samples

    def mutate(self, particles, beta, n_steps=None):
        raise NotImplementedError

    def log_prob(self, z, beta=None):
        """smc/base.py:507-519: tempered log-target in the preconditioned space, NaN (and +inf) -> -inf.
        Accepts numpy or torch input and answers in the same namespace (used by custom `mutate`
        implementations; the built-in mutation fuses this into the pCN kernel)."""
        x, log_abs_det_jacobian = self.preconditioning_transform.inverse(z)
        x_dev = self.engine.asarray(x, dtype=self.x_torch_dtype)
        log_q = self._flow_log_prob(x_dev)
        lp, ll = self._eval_prior_likelihood(x_dev, log_q)
        log_prob = (1 - beta) * log_q + beta * (ll + lp)
        if log_abs_det_jacobian is not None:
            log_prob = log_prob + self._to_dev(log_abs_det_jacobian)
        # NaN -> -inf (smc/base.py:518); +inf -> -inf as well, as every accept kernel maps it (csrc/asmc_pcn_dev.h, log_p_t)
        log_prob = torch.where(log_prob < math.inf, log_prob, torch.full_like(log_prob, -math.inf))
        return log_prob if is_torch(z) else to_numpy(log_prob)

    # ---- checkpoint glue (smc/base.py:521-562) -----------------------------------------------
    def build_checkpoint_state(self, samples: SMCSamples, iteration: int, beta: float) -> dict:
        """smc/base.py:521-530.  Sharded runs: every rank checkpoints ITS shard; the shard bookkeeping (rows per rank,
        population size, rank, world) travels in `meta` next to the reference's `beta`."""
        meta = {"beta": beta}
        if self.comm.sharded:
            meta.update(rank=self.comm.rank, world=self.comm.world, n_global=samples._n_global(),
                        shard_counts=samples.shard_counts_list())
        return super().build_checkpoint_state(samples.to_numpy(), iteration, meta=meta)

    def _checkpoint_extra_state(self) -> dict:
        """smc/base.py:532-544 (+ `pcn_state`: step size / step counter / nu of this repository's mutation kernel)."""
        rng_state = self.rng.bit_generator.state if hasattr(self.rng, "bit_generator") else None
        return {"history": copy.deepcopy(self.history), "rng_state": rng_state,
                "sampler_kwargs": getattr(self, "sampler_kwargs", None),
                "pcn_state": copy.deepcopy(getattr(self, "_pcn_state", None))}

    def restore_from_checkpoint(self, source):
        """smc/base.py:546-562."""
        samples, state = super().restore_from_checkpoint(source)
        meta = state.get("meta", {}) if isinstance(state, dict) else {}
        beta = meta.get("beta", None) if isinstance(meta, dict) else None
        if beta is None:
            beta = state.get("beta", 0.0)
        iteration = state.get("iteration", 0)
        self.history = state.get("history", SMCHistory())
        rng_state = state.get("rng_state")
        if rng_state is not None and hasattr(self.rng, "bit_generator"):
            self.rng.bit_generator.state = rng_state
        if state.get("pcn_state") is not None:
            self._pcn_state = state["pcn_state"]
        world = int(meta.get("world", 1)) if isinstance(meta, dict) else 1
        if world != self.comm.world or (world > 1 and int(meta.get("rank", 0)) != self.comm.rank):
            raise ValueError(f"checkpoint was written by rank {meta.get('rank', 0)} of {world}, this is rank "
                             f"{self.comm.rank} of {self.comm.world}: a sharded run resumes with the same layout")
        e = self.engine
        s = self._wrap(e.asarray(samples.x, dtype=self.x_torch_dtype), e.asarray(samples.log_likelihood),
                       e.asarray(samples.log_prior), e.asarray(samples.log_q), beta)
        if world > 1:
            s.n_global, s.shard_counts = int(meta["n_global"]), [int(c) for c in meta["shard_counts"]]
            s.ragged = len(set(s.shard_counts)) > 1
        return s, beta, iteration


class HipSMC(SMCSampler):
    """The `"smc"` / `"minipcn_smc"` sampler (smc/minipcn.py:14-135) with the fused pCN HIP kernel."""

    rng = None

    @track_calls
    def sample(self, n_samples: int, n_steps: int = None, min_beta_step: float | None = None,
               max_beta_step: float | None = None, max_n_steps: int | None = None, adaptive: bool = True,
               target_efficiency: float = 0.5, target_efficiency_rate: float = 1.0,
               n_final_samples: int | None = None, sampler_kwargs: dict | None = None, rng=None,
               checkpoint_callback=None, checkpoint_every: int | None = None,
               checkpoint_file_path: str | None = None, resume_from: str | bytes | dict | None = None,
               beta_tolerance: float = 1e-6, store_sample_history: bool = True,
               resample_mode: str | None = None, resample_method: str | None = None):
        self.sampler_kwargs = dict(sampler_kwargs or {})
        self.sampler_kwargs.setdefault("n_steps", 5 * self.dims)  # minipcn.py:46
        self.sampler_kwargs.setdefault("target_acceptance_rate", 0.234)  # minipcn.py:47
        # "tpcn" (t-preconditioned Crank-Nicolson, Student-t reference fitted per temperature) is the reference's
        # default (minipcn.py:48); "pcn" is the Gaussian-reference special case
        self.sampler_kwargs.setdefault("step_fn", "tpcn")
        self.sampler_kwargs.setdefault("verbose", False)
        # proposal-noise generator of the fused kernel: "f64" (fp64 Box-Muller, matches the oracle to 1e-12)
        # or "f32" (hardware fp32 Box-Muller, 4 normals per Philox block; the fast production mode)
        self.sampler_kwargs.setdefault("noise", "f64")
        if self.sampler_kwargs["step_fn"] not in ("pcn", "tpcn"):
            raise NotImplementedError(
                f"step_fn={self.sampler_kwargs['step_fn']!r} is not implemented by the HIP mutation kernels; "
                "use step_fn='tpcn' or 'pcn'")
        self.rng = rng or self.rng or np.random.default_rng()
        self._pcn_state = {"rho": None, "step": 0, "nu": None}
        return super().sample(
            n_samples, n_steps=n_steps, adaptive=adaptive, target_efficiency=target_efficiency,
            target_efficiency_rate=target_efficiency_rate, n_final_samples=n_final_samples,
            min_beta_step=min_beta_step, max_beta_step=max_beta_step, max_n_steps=max_n_steps,
            checkpoint_callback=checkpoint_callback, checkpoint_every=checkpoint_every,
            checkpoint_file_path=checkpoint_file_path, resume_from=resume_from, beta_tolerance=beta_tolerance,
            store_sample_history=store_sample_history, resample_mode=resample_mode,
            resample_method=resample_method)

    # ---- reference Gaussian of the pCN proposal ----------------------------------------------
    def _speculated_moments_n(self, samples) -> int | None:
        """The population size when the coming mutation fits its reference Gaussian to the moments of the whole resampled
        population in x space (`pcn` steps, identity preconditioning): the fused importance step (one rank) or the sharded
        step's finish then computes them right behind its gather.  None otherwise."""
        T = self.preconditioning_transform
        if ((self.comm.sharded and not (self.shard_layout == "owner" and smc_math.shard_step_available(self.engine, self.comm)))
                or self.sampler_kwargs.get("step_fn", "tpcn") != "pcn"
                or not (isinstance(T, IdentityTransform) or getattr(T, "is_identity", False))):
            return None
        return int(self._n_global(samples))

    def _factor_ahead(self) -> bool:
        """The reference Gaussian is factored on the device (`_fit_reference_gaussian`'s device fit): the speculated importance
        step may enqueue the factorisation right behind the moments (ASMC_FACTOR_AHEAD=0: A-B switch)."""
        return (hasattr(self.engine, "reference_factor") and self.dims <= 128 and not os.environ.get("ASMC_HOST_REFERENCE_FIT")
                and os.environ.get("ASMC_FACTOR_AHEAD", "1") != "0")

    def _resample_moments_n(self, samples) -> int | None:
        T = self.preconditioning_transform
        if (self.sampler_kwargs.get("step_fn", "tpcn") != "pcn"
                or not (isinstance(T, IdentityTransform) or getattr(T, "is_identity", False))):
            return None
        return int(self._n_global(samples))

    def _fit_reference_gaussian(self, x: torch.Tensor, n_global: int | None = None, moments=None):
        """Population mean and covariance (ddof=1) over ALL ranks -> (mu, L, Linv) on device.  `moments`: what the fused
        importance step parked for exactly these rows (`SMCSamples.speculate_importance_step`)."""
        e, comm = self.engine, self.comm
        self.__dict__.pop("_ref_fit_generation", None)  # (set below / by _fit_reference for the request THIS fit makes or takes over)
        n = n_global or x.shape[0] * comm.world
        device_fit = hasattr(e, "reference_factor") and x.shape[1] <= 128 and not os.environ.get("ASMC_HOST_REFERENCE_FIT")
        if (moments is not None and moments[0] == x.data_ptr() and moments[1] == tuple(x.shape)
                and moments[2] == n and moments[3] == getattr(e, "_gram_gen", None)):
            if device_fit:  # mean, covariance, Cholesky factor and its inverse never visit the host (asmc_reference_factor)
                self._ref_fit_pending = True
                if len(moments) > 4 and moments[4] is not None:
                    # factored already, right behind the moments (speculate_importance_step's factor_ahead)
                    self._ref_fit_generation = moments[4][3]
                    return moments[4][:3]
                return e.reference_factor(x.shape[1], n, n)
            s, g = e.mean_gram_fetch(x.shape[1])  # enqueued behind the importance step's gather; waits for the stream
            mean = s / n
        elif not comm.sharded and hasattr(e, "mean_gram"):
            if device_fit and e.mean_gram_enqueue(x, n):
                self._ref_fit_pending = True
                return e.reference_factor(x.shape[1], n, n)
            s, g = e.mean_gram(x, n)  # both passes enqueued together: the centre never visits the host
            mean = s / n
        elif hasattr(e, "mean_gram_across_ranks_ok") and e.mean_gram_across_ranks_ok(x, comm):
            if device_fit and e.mean_gram_enqueue(x, n, comm):
                self._ref_fit_pending = True
                return e.reference_factor(x.shape[1], n, n)
            s, g = e.mean_gram(x, n, comm)  # ... and both sums cross the ranks on the stream (RCCL all-reduce)
            mean = s / n
        elif (device_fit and comm.sharded and hasattr(e, "colsum_dev") and hasattr(comm, "all_reduce_sum_")
              and x.shape[1] in (32, 64, 128) and not os.environ.get("ASMC_GRAM_GENERIC")):
            # sharded, no communicator of the library's own: the rank's sums and Gram matrix stay on the device, torch.distributed
            # sums them over the ranks on the stream (RCCL), the factorisation follows - no host round trip
            xs = x if x.data_ptr() % 16 == 0 else x.clone()  # (a rank-local property must not pick the code path of a sharded run)
            s_d = comm.all_reduce_sum_(e.colsum_dev(xs))
            g_d = comm.all_reduce_sum_(e.centered_gram_dev(xs, s_d, n))
            self._ref_fit_pending = True
            return e.reference_factor(x.shape[1], n, n, moments_dev=(s_d, g_d))
        else:
            parts = comm.all_gather_f64(e.colsum(x))
            s = parts[0].copy()
            for r in range(1, comm.world):
                s = s + parts[r]
            mean = s / n
            parts = comm.all_gather_f64(e.centered_gram(x, mean))
            g = parts[0].copy()
            for r in range(1, comm.world):
                g = g + parts[r]
        if device_fit:  # the same factorisation kernel on every path of a device engine: the same bits single-rank and sharded
            self._ref_fit_pending = True
            return e.reference_factor(x.shape[1], n, n, moments=(s, g))
        cov = g / max(n - 1, 1)
        cov = 0.5 * (cov + cov.T)
        scale = float(np.mean(np.diag(cov)))
        if not np.isfinite(scale) or scale <= 0:
            scale = 1.0
        # The d x d factorisation is control-plane scalar work on the host.  It must NOT wake a multi-threaded
        # BLAS pool: on the GPU box (256 host cores) OpenBLAS' spinning worker threads stalled the HIP runtime
        # for ~60 ms after every few calls (measured; tools/smcprof.py), tripling the sampler's wall time.
        with _single_threaded_blas():
            jitter = 0.0
            for _ in range(12):
                try:
                    L = np.linalg.cholesky(cov + jitter * scale * np.eye(self.dims))
                    break
                except np.linalg.LinAlgError:
                    jitter = 1e-12 if jitter == 0.0 else jitter * 100
            else:
                raise RuntimeError("could not factor the particle covariance")
            Linv = np.linalg.inv(L)
        return self._upload_reference(mean, L, Linv)

    def _upload_reference(self, mean, L, Linv):
        """(mu, tril L, tril Linv) on the device through ONE upload: each pageable host-to-device copy costs ~30 us of
        idle GPU at a temperature boundary."""
        d = self.dims
        seg = -(-d // 32) * 32  # segment starts stay 256-byte aligned
        size = seg + 2 * seg * d
        device = getattr(self.engine, "device", None)
        if isinstance(device, torch.device) and device.type == "cuda":
            # pinned staging + one asynchronous copy on the stream (a pageable upload is a blocking ~40 us); both buffers are
            # reused: the previous temperature's kernels have finished with them when its mutation's results were collected
            bufs = self.__dict__.get("_ref_bufs")
            if bufs is None or bufs[0].numel() != size:
                bufs = self._ref_bufs = (torch.zeros(size, dtype=torch.float64).pin_memory(),
                                         torch.empty(size, dtype=torch.float64, device=device))
            host_t, dev = bufs
            host = host_t.numpy()
        else:
            host_t, dev, host = None, None, np.zeros(size)
        host[:d] = mean
        host[seg:seg + d * d] = np.tril(L).reshape(-1)
        host[seg + seg * d:seg + seg * d + d * d] = np.tril(Linv).reshape(-1)
        if dev is None:
            dev = self.engine.asarray(host)
        else:
            dev.copy_(host_t, non_blocking=True)
        return dev[:d], dev[seg:seg + d * d].view(d, d), dev[seg + seg * d:seg + seg * d + d * d].view(d, d)

    def _fit_reference(self, x: torch.Tensor, n_global: int, step_fn: str, moments=None):
        """(mu, L, Linv, nu) of the mutation's reference distribution: Gaussian moments of the whole population
        (`pcn`, nu = 0) or a Student-t fitted by EM to a strided subsample of `tpcn_fit_subsample` particles, the
        same on every rank (`tpcn`; student_t.py).  A fit with nu above NU_GAUSSIAN runs the Gaussian kernels."""
        if step_fn != "tpcn":
            out = self._fit_reference_gaussian(x, n_global, moments)
            if self.__dict__.get("_ref_fit_pending") and self.__dict__.get("_ref_fit_generation") is None:
                # the request just made (later ones - the next temperature's, enqueued behind this mutation - get their own cell)
                self._ref_fit_generation = getattr(self.engine, "ref_generation", None)
            return (*out, 0.0)
        from ..student_t import NU_GAUSSIAN, _chol, fit_student_t_device

        e, comm = self.engine, self.comm
        # The fit adapts the kernel to the particles it then moves, which biases log Z by O(parameters / fitted rows): with
        # 2048 rows the d = 128 runs of BASELINE config 5 came out +0.9 sigma high on average over 30 seeds (rms 1.26), with
        # d^2 rows -0.15 sigma (rms 0.78) for 4 % more wall time; d = 32 shows nothing at 2048 (tools/validate_logz.py,
        # profiles/r02_tpcn_subsample_bias.txt).
        m = min(int(self.sampler_kwargs.get("tpcn_fit_subsample", max(2048, self.dims * self.dims))), 16384)
        k = max(1, min(m, n_global) // comm.world)
        n_local = x.shape[0]
        rows = torch.as_tensor((np.arange(k, dtype=np.int64) * n_local) // k, device=x.device)
        sub = x[rows].to(torch.float64).contiguous()  # strided subsample, stays on the device
        if comm.sharded:
            sub = comm.all_gather_tensor(sub)  # the same rows on every rank: identical fits
        st = self._pcn_state
        # EM sweeps: a cold start needs ~a dozen; later temperatures restart (mu, Sigma) from the subsample's
        # moments and nu from the previous fit, and a few sweeps track the slowly changing population
        iters = int(self.sampler_kwargs.get("tpcn_fit_iters", 12 if st.get("nu") is None else 4))
        fit = None
        # every sweep on the stream, one synchronisation (config 5, d = 128: 0.87 s per run against 0.88-0.97 s with the host's
        # LAPACK and its round trips, which depend on the box's host)
        if (hasattr(e, "student_fit") and self.dims <= int(os.environ.get("ASMC_DEVICE_EM_MAX_D", 128))
                and not os.environ.get("ASMC_HOST_REFERENCE_FIT")):
            fit = e.student_fit(sub, iters, 1e-3, st.get("nu") or 20.0)
        if fit is not None:
            (mu_d, L_d, Linv_d), nu, _, status, _, _ = fit
            if status < 0:
                raise RuntimeError("could not factor the scale matrix of the Student-t fit")
            st["nu"] = nu
            self.history.mcmc_nu.append(float(nu) if nu <= NU_GAUSSIAN else float("inf"))
            return mu_d, L_d, Linv_d, (nu if nu <= NU_GAUSSIAN else 0.0)
        with _single_threaded_blas():
            mean, cov, nu = fit_student_t_device(e, sub, max_iter=iters, nu0=st.get("nu") or 20.0)
            L = _chol(cov)
            Linv = np.linalg.inv(L)
        st["nu"] = nu
        self.history.mcmc_nu.append(float(nu) if nu <= NU_GAUSSIAN else float("inf"))
        return (*self._upload_reference(mean, L, Linv), (nu if nu <= NU_GAUSSIAN else 0.0))

    def _mutate_preconditioned(self, particles: SMCSamples, x: torch.Tensor, beta: float, n_steps: int, target: float):
        """smc/minipcn.py:105-132 with a non-trivial preconditioning transform: the chain runs in z = T(x) (refit at
        every temperature), the tempered log-target there is log p_t(T^-1(z)) + log|det dT^-1/dz| (smc/base.py:507-519).
        Per step: propose in z -> x', log|J| (asmc_transform_inverse) -> densities at x' -> accept; the carried
        log-Jacobian follows the accepted state inside the accept kernel."""
        e, comm, T = self.engine, self.comm, self.preconditioning_transform
        if getattr(T, "engine", None) is None and hasattr(T, "engine"):
            T.engine = e
        ll, lp, lq = particles.log_likelihood, particles.log_prior, particles.log_q
        n_local = x.shape[0]
        n_global = self._n_global(particles)
        gid0 = self._gid0(particles)
        try:
            z = T.fit(x, comm=comm)
        except TypeError:  # a user-supplied transform with the reference's fit(x) signature
            z = T.fit(x)
        z = e.asarray(z, dtype=x.dtype)
        logj = e.asarray(T.inverse(z)[1])
        mu, L, Linv, nu = self._fit_reference(z, n_global, self.sampler_kwargs.get("step_fn", "tpcn"))
        st = self._pcn_state
        if st["rho"] is None:
            st["rho"] = min(2.38 / math.sqrt(self.dims), 0.99)
        seed = int(self.rng.integers(0, 2**63 - 1, dtype=np.int64))
        step0 = st["step"]
        acc_rates = []

        # log q(x') straight from z' when the flow's data transform and T share their bounded stage (no erfinv round trip)
        logq_z = None
        if self.sampler_kwargs.get("flow_from_preconditioned", True) and hasattr(self.prior_flow, "log_prob_from_preconditioned"):
            logq_z = self.prior_flow.log_prob_from_preconditioned(T)

        def flow_lq(z_prop, x_prop, logj_new):
            nonlocal logq_z
            if logq_z is not None and z_prop.dtype in (torch.float64, torch.float32) and z_prop.is_contiguous():
                try:
                    return logq_z(z_prop, logj_new)
                except Exception as exc:  # a row shape the premapped kernel does not take: the round trip from here on
                    logger.info("log q from the preconditioned coordinate is not available: %s", exc)
                    logq_z = None
            return self._flow_log_prob(x_prop)

        def step(t, rho):
            z_prop, q0, q1 = e.pcn_propose(z, mu, L, Linv, rho, seed, gid0, step0 + t, nu=nu)
            x_prop, logj_new = T.inverse(z_prop)
            x_prop, logj_new = e.asarray(x_prop, dtype=x.dtype), e.asarray(logj_new)
            lq_new = flow_lq(z_prop, x_prop, logj_new)
            lp_new, ll_new = self._eval_prior_likelihood(x_prop, lq_new)
            return z_prop, q0, q1, ll_new, lp_new, lq_new, logj_new

        if hasattr(e, "pcn_split_begin"):
            # step size and accept counts stay on the device: the host enqueues step t + 1 while step t runs
            if comm.sharded:
                e.set_count_hook(comm, n_global)
            try:
                done = 0
                while done < n_steps and hasattr(e, "pcn_ysplit_begin"):
                    # whitened-state session on z (d = 4, 8, 16, 32): one mat-vec per step, LDS-free accept; the carried
                    # log-Jacobian follows the accepted state inside the accept kernel
                    chunk = min(n_steps - done, 2048)
                    sess = e.pcn_ysplit_begin(z, beta, mu, L, Linv, seed, gid0, st["rho"], target, True, nu,
                                              self.sampler_kwargs.get("noise", "f64"))
                    if sess is None:
                        break
                    # inverse transform (and log q, when it is available from z') inside the propose kernel where the
                    # transform has one bounded stage and nothing periodic
                    t_dev = None
                    if (self.sampler_kwargs.get("fuse_transform", True) and hasattr(e, "pcn_ysplit_propose_tr")
                            and hasattr(T, "_tables") and not np.any(getattr(T, "_periodic", [1]))):
                        td = T._tables()[1]
                        if td.hints & 7 in (5, 3):  # ASMC_TR_NO_PERIODIC | NO_PROBIT (logit) / NO_LOGIT (probit)
                            t_dev = td
                    for t in range(done, done + chunk):
                        if t_dev is not None:
                            x_prop, logj_new, lq_new = e.pcn_ysplit_propose_tr(sess, step0 + t, t_dev,
                                                                               getattr(logq_z, "fused_args", None))
                            if lq_new is None:
                                lq_new = self._flow_log_prob(x_prop)
                        else:
                            z_prop = e.pcn_ysplit_propose(sess, step0 + t)
                            x_prop, logj_new = T.inverse(z_prop)
                            x_prop, logj_new = e.asarray(x_prop, dtype=x.dtype), e.asarray(logj_new)
                            lq_new = flow_lq(z_prop, x_prop, logj_new)
                        lp_new, ll_new = self._eval_prior_likelihood(x_prop, lq_new)
                        e.pcn_ysplit_accept(sess, step0 + t, ll, lp, lq, ll_new, lp_new, lq_new, n_global, t - done,
                                            logj=logj, logj_new=logj_new)
                    n_acc, _, st["rho"] = e.pcn_ysplit_end(sess, chunk)
                    acc_rates.extend((n_acc / n_global).tolist())
                    done += chunk
                while done < n_steps:
                    chunk = min(n_steps - done, 2048)
                    e.pcn_split_begin(st["rho"])
                    for t in range(done, done + chunk):
                        z_prop, q0, q1, ll_new, lp_new, lq_new, logj_new = step(t, 0.0)
                        e.pcn_accept(z, z_prop, ll, lp, lq, ll_new, lp_new, lq_new, q0, q1, beta, seed, gid0, step0 + t,
                                     logj_old=logj, logj_new=logj_new, want_count=False)
                        e.pcn_split_adapt(n_global, target, t - done, True)
                    n_acc, _, st["rho"] = e.pcn_split_end(chunk)
                    acc_rates.extend((n_acc / n_global).tolist())
                    done += chunk
            finally:
                if comm.sharded:
                    e.set_count_hook(None, None)
        else:
            for t in range(n_steps):
                z_prop, q0, q1, ll_new, lp_new, lq_new, logj_new = step(t, st["rho"])
                n_acc = e.pcn_accept(z, z_prop, ll, lp, lq, ll_new, lp_new, lq_new, q0, q1, beta, seed, gid0, step0 + t,
                                     logj_old=logj, logj_new=logj_new)
                tot = float(comm.all_gather_f64(np.array([float(n_acc)])).sum())
                acc_rates.append(tot / n_global)
                st["rho"] = pcn_adapt(st["rho"], acc_rates[-1], target, t)
        st["step"] = step0 + n_steps
        self.history.mcmc_acceptance.append(float(np.mean(acc_rates)))
        self.history.mcmc_step_size.append(float(st["rho"]))
        x_new = e.asarray(T.inverse(z)[0], dtype=x.dtype)
        if self._global_counts([e.count_nonfinite(lq)[0]])[0]:
            raise ValueError("Log proposal contains NaN values")
        return self._wrap(x_new, ll, lp, lq, beta, like=particles)

    def _adapt_mode(self) -> int:
        """`asmc_pcn_params.adapt` of the device-side step loops: 1 = the step size adapts after every step (default, the
        specification every test pins); `sampler_kwargs["adapt_lag"] = k` (2 .. 64) holds it for blocks of k steps and applies the
        block's k updates at its end, in order, each with its own step's count - a sharded run then exchanges accept counts once per
        block (n_steps / k all-reduces per temperature instead of n_steps).  Single-rank and sharded runs with the same k agree bit
        for bit; runs with different k are different (equally valid) adaptation schedules.  The callables / preconditioned paths
        adapt after every step whatever the value."""
        k = int(self.sampler_kwargs.get("adapt_lag", 1) or 1)
        if not 1 <= k <= 64:
            raise ValueError(f"adapt_lag must be in 1 .. 64, got {k}")
        return k

    def _device_flow(self):
        """The proposal flow packed for the MFMA kernel, or None (not a float32 coupling flow of a supported shape)."""
        if not hasattr(self.prior_flow, "device_coupling"):
            return self._adapted_zuko_flow()
        if not hasattr(self.engine, "coupling_logprob"):  # (as `_flow_log_prob`: an engine without flow kernels - the tests' CPU double)
            return None
        try:
            return self.prior_flow.device_coupling(self.engine)
        except (ValueError, RuntimeError) as exc:
            logger.info("flow stays on its torch modules: %s", exc)
            return None

    def _adapted_zuko_flow(self):
        """A proposal that is the reference's own `ZukoFlow(flow_class="MAF")` (what `Aspire.fit` trains and hands to the sampler
        through the plug-in seam, flows/torch/flows.py:156-168) has no `device_coupling`; its zuko module's state dict is repacked
        for the HIP kernels by `MAFFlow.from_zuko_state_dict`, so that the mutation runs the one-kernel flow step instead of the
        callables split path.  zuko is absent from the build image: the adapter follows zuko's documented layout and is UNVERIFIED
        against the package, so every adapted flow is cross-checked against the flow's OWN `log_prob` before it is used - on 256
        standard-normal points and 256 of the flow's own draws - and declined on a mismatch (the flow then stays on its modules).
        On by default since round 6 (the check decides); `sampler_kwargs["zuko_adapter"] = False` switches it off."""
        if not self.sampler_kwargs.get("zuko_adapter", True) or not hasattr(self.engine, "coupling_logprob"):
            return None
        pf = self.prior_flow
        inner = getattr(pf, "_flow", None)
        if inner is None or not hasattr(inner, "state_dict"):
            return None
        dt = getattr(pf, "data_transform", None)
        if dt is not None and type(dt).__name__ != "IdentityTransform":
            logger.info("zuko adapter: the flow lives behind a data transform; it stays on its own modules")
            return None
        cache = self.__dict__.get("_zuko_cache")
        # keyed on the parameters' in-place version counters and storage, not on the module's identity alone: a second fit()
        # retrains the SAME module object in place, and id() values are reused after garbage collection
        try:
            key = (id(inner), tuple((p.data_ptr(), int(p._version)) for p in inner.parameters()),
                   tuple((b.data_ptr(), int(b._version)) for b in inner.buffers()))
        except Exception:
            key = object()  # cannot tell whether the flow changed: rebuild (and cross-check) on every call
        if cache is None or cache[0] != key:
            from ..flows import MAFFlow

            try:
                adapted = MAFFlow.from_zuko_state_dict(inner.state_dict(), device=self.engine.device)
                dev = adapted.device_coupling(self.engine)
                probe = torch.randn((256, self.dims), device=self.engine.device, dtype=torch.float64)
                try:  # ... and where the flow itself puts its mass
                    own = pf.sample_and_log_prob(256)[0]
                    own = self._to_dev(own).reshape(256, self.dims).to(torch.float64)
                    probe = torch.cat([probe, own[torch.isfinite(own).all(dim=1)]])
                except Exception as exc:
                    logger.info("zuko adapter: no draws from the flow for the cross-check (%s); standard-normal probes only", exc)
                mine = self.engine.coupling_logprob(probe, dev)
                theirs = self._to_dev(pf.log_prob(probe.to(getattr(pf, "dtype", torch.float32))))
                err = float(((mine - theirs).abs() / theirs.abs().clamp_min(1.0)).max())
                if not err <= 1e-4:
                    raise ValueError(f"adapted flow disagrees with the flow's own log_prob (max relative difference {err:.3g})")
                logger.info("zuko adapter in use (unverified against zuko itself; agrees with this flow's log_prob to %.1e on %d points)", err, len(probe))
            except Exception as exc:
                logger.warning("zuko adapter declined: %s", exc)
                dev = None
            cache = self._zuko_cache = (key, dev)
        return cache[1]

    def _flow_fused_ok(self, dev_flow) -> bool:
        return (dev_flow is not None and isinstance(self._log_likelihood, DiagGaussianMixture)
                and isinstance(self._log_prior, DiagGaussianMixture)
                and isinstance(self.preconditioning_transform, IdentityTransform))

    def _fused_ok(self) -> bool:
        return (isinstance(self._log_likelihood, DiagGaussianMixture)
                and isinstance(self._log_prior, DiagGaussianMixture)
                and hasattr(self.prior_flow, "device_mixture")
                and not getattr(self.prior_flow, "_has_transform", lambda: False)()  # log q is Gaussian in x itself
                and isinstance(self.preconditioning_transform, IdentityTransform))

    def mutate(self, particles: SMCSamples, beta: float, n_steps: int | None = None) -> SMCSamples:
        """smc/minipcn.py:69-135."""
        e, comm = self.engine, self.comm
        kwargs = self.sampler_kwargs.copy()
        n_steps = n_steps or kwargs.pop("n_steps")
        target = float(kwargs.get("target_acceptance_rate", 0.234))
        noise = kwargs.get("noise", "f64")
        x = particles.x if particles.x.is_contiguous() else particles.x.contiguous()
        T = self.preconditioning_transform
        transformed = not (isinstance(T, IdentityTransform) or getattr(T, "is_identity", False))
        if transformed:
            self.last_mutation_path = "preconditioned chain (z = T(x)): split propose / accept around the transform"
            out = self._mutate_preconditioned(particles, x, beta, n_steps, target)
            self._check_reference_fit()
            return out
        self.fit_preconditioning_transform(particles.x)
        ll, lp, lq = particles.log_likelihood, particles.log_prior, particles.log_q
        n_local = x.shape[0]
        n_global = self._n_global(particles)
        gid0 = self._gid0(particles)
        mu, L, Linv, nu = self._fit_reference(x, n_global, kwargs.get("step_fn", "tpcn"), particles.__dict__.get("_moments"))
        st = self._pcn_state
        if st["rho"] is None:
            st["rho"] = min(2.38 / math.sqrt(self.dims), 0.99)
        seed = int(self.rng.integers(0, 2**63 - 1, dtype=np.int64))
        step0 = st["step"]
        acc_rates = []
        dev_flow = self._device_flow()
        # whole step loop on the device: always for one rank; sharded when the engine can exchange the accept counts
        # between a step and its adaptation on the stream (asmc_pcn_set_count_hook), else one host round trip per step
        on_device = not comm.sharded or hasattr(e, "set_count_hook")
        if comm.sharded and on_device:
            e.set_count_hook(comm, n_global)
        try:
            out = self._mutate_steps(particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0,
                                     acc_rates, dev_flow, on_device, n_local, n_global, gid0, nu)
        finally:
            if comm.sharded and on_device:
                e.set_count_hook(None, None)
        self._check_reference_fit()
        return out

    def _check_reference_fit(self):
        """The device-side factorisation of the reference covariance (`engine.reference_factor`) reports behind the mutation
        it served: the steps' results have been collected, so its status is on the host."""
        if not self.__dict__.pop("_ref_fit_pending", False):
            return
        # the request that served THIS mutation: the next temperature's factorisation may already sit on the stream behind it
        gen = self.__dict__.pop("_ref_fit_generation", None)
        status = self.engine.reference_factor_status(gen) if gen else self.engine.reference_factor_status()
        if status == -2:  # nothing has synchronised the stream since (a mutation of zero steps)
            torch.cuda.synchronize(self.engine.device)
            status = self.engine.reference_factor_status(gen) if gen else self.engine.reference_factor_status()
        if status < 0:
            raise RuntimeError("could not factor the particle covariance")
        if status > 0:
            logger.info("reference covariance factored with jitter (try %d)", status)

    def _mutate_steps(self, particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates,
                      dev_flow, on_device, n_local, n_global, gid0, nu=0.0):
        """The mutation's step loop.  Four code paths, chosen from what the densities are (`last_mutation_path` names the one
        taken): the flow-proposal loop on the device, the built-in-density loop on the device, arbitrary callables between
        device-side propose / accept halves, and the same with one host round trip per step."""
        e = self.engine
        m = (particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates, dev_flow, on_device,
             n_local, n_global, gid0, nu)
        if self._flow_fused_ok(dev_flow):
            self._steps_flow_on_device(m)
        elif self._fused_ok():
            self._steps_builtin_on_device(m)
        elif hasattr(e, "pcn_split_begin") and on_device:
            self._steps_callables_split(m)
        else:
            self._steps_callables_round_trip(m)
        return self._finish_mutation(m)

    def _steps_flow_on_device(self, m):
        """Coupling-flow proposal density evaluated on the matrix cores inside the device-side step loop (BASELINE config 3)."""
        e, comm = self.engine, self.comm
        (particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates, dev_flow, on_device,
         n_local, n_global, gid0, nu) = m
        # flow proposal density evaluated on the MFMA inside the device-side step loop (BASELINE config 3)
        self.last_mutation_path = "flow: device-side step loop (asmc_pcn_mutate_flow; one fused kernel per step where its shape is covered)"
        t_ll = self._log_likelihood.device_mixture(e)
        t_lp = self._log_prior.device_mixture(e)
        if on_device:
            done = 0
            while done < n_steps:
                chunk = min(n_steps - done, 2048)
                ahead = chunk == n_steps and n_steps <= 1024 and self._importance_step_follows(beta)
                if ahead:
                    # the next temperature's importance step (search, resampling, gather, moments) goes onto the stream
                    # right behind the mutation, BEFORE the host waits for either: the GPU does not idle while Python
                    # does the bookkeeping of the finished mutation
                    handle = e.pcn_mutate_flow_enqueue(x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, dev_flow, seed, gid0,
                                                       st["rho"], chunk, step0, target, self._adapt_mode(), noise, nu)
                    out = self._wrap(x, ll, lp, lq, beta, like=particles)
                    ok = out.speculate_importance_step(self.current_target_efficiency(beta), self._beta_tolerance, self.rng,
                                                       resample_mode=self.resample_mode, resample_method=self.resample_method,
                                                       moments_n=self._speculated_moments_n(out), defer=True,
                                                       shard_layout=self.shard_layout, factor_ahead=self._factor_ahead())
                    n_acc, rho_hist, rho = e.pcn_mutate_flow_result(handle)  # waits for the mutation only
                    if ok:
                        st["prewrapped"] = out  # the step's results are collected when the next iteration asks for them
                else:
                    n_acc, rho_hist, rho = e.pcn_mutate_flow(x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, dev_flow, seed,
                                                             gid0, st["rho"], chunk, step0 + done, target, self._adapt_mode(), noise, nu)
                st["rho"] = rho
                st["lq_checked"] = hasattr(e, "pcn_lq_nan")
                acc_rates.extend((n_acc / n_global).tolist())
                done += chunk
                n_bad = e.pcn_flow_nonfinite() if hasattr(e, "pcn_flow_nonfinite") else 0
                if n_bad > 0:
                    logger.warning(f"{n_bad} proposals had a non-finite flow density and were rejected (a badly scaled flow "
                                   "overflows the fp16 operand pairs of the flow kernel; ASMC_FLOW_MATH=f32 uses fp32 MFMAs)")
        else:
            for t in range(n_steps):
                n_acc, _, _ = e.pcn_mutate_flow(x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, dev_flow, seed, gid0,
                                                st["rho"], 1, step0 + t, target, False, noise, nu)
                tot = float(comm.all_gather_f64(np.array([float(n_acc[0])])).sum())
                acc_rates.append(tot / n_global)
                st["rho"] = pcn_adapt(st["rho"], acc_rates[-1], target, t)
        self.n_likelihood_evaluations += n_steps * n_local

    def _steps_builtin_on_device(self, m):
        """Built-in densities (DiagGaussianMixture targets, analytic proposal): the whole loop in asmc_pcn_mutate."""
        e, comm = self.engine, self.comm
        (particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates, dev_flow, on_device,
         n_local, n_global, gid0, nu) = m
        self.last_mutation_path = "built-in densities: device-side step loop (asmc_pcn_mutate)"
        t_ll = self._log_likelihood.device_mixture(e)
        t_lp = self._log_prior.device_mixture(e)
        t_lq = self.prior_flow.device_mixture(e)
        if on_device:
            done = 0
            while done < n_steps:
                chunk = min(n_steps - done, 2048)
                n_acc, rho_hist, rho = e.pcn_mutate(x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, t_lq, seed,
                                                    gid0, st["rho"], chunk, step0 + done, target, self._adapt_mode(), noise, nu)
                st["rho"] = rho
                st["lq_checked"] = hasattr(e, "pcn_lq_nan")
                acc_rates.extend((n_acc / n_global).tolist())
                done += chunk
        else:
            for t in range(n_steps):
                n_acc, _, _ = e.pcn_mutate(x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, t_lq, seed, gid0,
                                           st["rho"], 1, step0 + t, target, False, noise, nu)
                tot = float(comm.all_gather_f64(np.array([float(n_acc[0])])).sum())
                acc_rates.append(tot / n_global)
                st["rho"] = pcn_adapt(st["rho"], acc_rates[-1], target, t)
        self.n_likelihood_evaluations += n_steps * n_local

    def _steps_callables_split(self, m):
        """Arbitrary callables between the propose and accept halves; step size and accept counts stay on the device."""
        e, comm = self.engine, self.comm
        (particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates, dev_flow, on_device,
         n_local, n_global, gid0, nu) = m
        # arbitrary callables between propose and accept; step size and accept counts stay on the device (the exchange
        # hook of sharded runs is already installed by the caller), so the host enqueues step t + 1 while step t runs
        done = 0
        self.last_mutation_path = "callables: split propose / accept with the step closed on the stream (asmc_pcn_split_*)"
        while done < n_steps and hasattr(e, "pcn_ysplit_begin"):
            # whitened-state session (d = 4, 8, 16, 32): the chain state stays coordinate-major on the device, a step is
            # one mat-vec in the propose kernel and an LDS-free accept kernel
            chunk = min(n_steps - done, 2048)
            sess = e.pcn_ysplit_begin(x, beta, mu, L, Linv, seed, gid0, st["rho"], target, True, nu, noise)
            if sess is None:
                break
            self.last_mutation_path = "callables: whitened-state session (asmc_pcn_ysplit_*)"
            for t in range(done, done + chunk):
                x_prop = e.pcn_ysplit_propose(sess, step0 + t)
                lq_new = self._flow_log_prob(x_prop)
                lp_new, ll_new = self._eval_prior_likelihood(x_prop, lq_new)
                e.pcn_ysplit_accept(sess, step0 + t, ll, lp, lq, ll_new, lp_new, lq_new, n_global, t - done)
            n_acc, _, st["rho"] = e.pcn_ysplit_end(sess, chunk)
            acc_rates.extend((n_acc / n_global).tolist())
            done += chunk
        while done < n_steps:
            chunk = min(n_steps - done, 2048)
            e.pcn_split_begin(st["rho"])
            for t in range(done, done + chunk):
                x_prop, q0, q1 = e.pcn_propose(x, mu, L, Linv, 0.0, seed, gid0, step0 + t, nu=nu)
                lq_new = self._flow_log_prob(x_prop)
                lp_new, ll_new = self._eval_prior_likelihood(x_prop, lq_new)
                e.pcn_accept(x, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, q0, q1, beta, seed, gid0, step0 + t,
                             want_count=False)
                e.pcn_split_adapt(n_global, target, t - done, True)
            n_acc, _, st["rho"] = e.pcn_split_end(chunk)
            acc_rates.extend((n_acc / n_global).tolist())
            done += chunk

    def _steps_callables_round_trip(self, m):
        """Arbitrary callables, one host round trip per step (engines without the split entry points; sharded without a count hook)."""
        e, comm = self.engine, self.comm
        (particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates, dev_flow, on_device,
         n_local, n_global, gid0, nu) = m
        self.last_mutation_path = "callables: split propose / accept, one host round trip per step"
        for t in range(n_steps):
            x_prop, q0, q1 = e.pcn_propose(x, mu, L, Linv, st["rho"], seed, gid0, step0 + t, nu=nu)
            lq_new = self._flow_log_prob(x_prop)
            lp_new, ll_new = self._eval_prior_likelihood(x_prop, lq_new)
            n_acc = e.pcn_accept(x, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, q0, q1, beta, seed, gid0,
                                 step0 + t)
            tot = float(comm.all_gather_f64(np.array([float(n_acc)])).sum())
            acc_rates.append(tot / n_global)
            st["rho"] = pcn_adapt(st["rho"], acc_rates[-1], target, t)

    def _finish_mutation(self, m):
        """Bookkeeping behind every path: step counter, history, the NaN check of the carried log q, the wrapped population."""
        e, comm = self.engine, self.comm
        (particles, x, ll, lp, lq, beta, n_steps, target, noise, mu, L, Linv, st, seed, step0, acc_rates, dev_flow, on_device,
         n_local, n_global, gid0, nu) = m
        st["step"] = step0 + n_steps
        self.history.mcmc_acceptance.append(float(np.mean(acc_rates)))
        self.history.mcmc_step_size.append(float(st["rho"]))
        # the device-side loops count the NaNs of the carried log q themselves and return the count with their results
        n_nan = e.pcn_lq_nan() if st.pop("lq_checked", False) else e.count_nonfinite(lq)[0]
        # Sharded runs: every rank must take the same decision, which costs a collective and a synchronisation per
        # temperature.  While another importance step follows, its beta search counts the NaN weights of ALL ranks in the
        # records it exchanges anyway and raises there ("Log weights contain NaN values"); only the last mutation asks the ranks.
        deferred = comm.sharded and beta < 1.0 and self.adaptive and self.device_bisection
        pre = st.pop("prewrapped", None)
        if not deferred and self._global_counts([n_nan])[0]:
            raise ValueError("Log proposal contains NaN values")
        return pre if pre is not None else self._wrap(x, ll, lp, lq, beta, like=particles)


_BLAS_CONTROLLER = []  # the process's ThreadpoolController, discovered once (the discovery walks every loaded library: ~1 ms)


def _single_threaded_blas():
    try:
        if not _BLAS_CONTROLLER:
            from threadpoolctl import ThreadpoolController

            _BLAS_CONTROLLER.append(ThreadpoolController())
        return _BLAS_CONTROLLER[0].limit(limits=1)
    except Exception:  # threadpoolctl missing or too old: fall through (only a performance matter)
        import contextlib

        return contextlib.nullcontext()


def pcn_adapt(rho: float, acc: float, target: float, t: int) -> float:
    """Step-size adaptation of this repository's pCN spec, identical to the device
    version in k_pcn_adapt: log rho += (acc - target)/(t+1)^0.75, rho clipped to [1e-4, 0.99]."""
    r = math.exp(math.log(rho) + (acc - target) / (t + 1) ** 0.75)
    return min(max(r, 1e-4), 0.99)
