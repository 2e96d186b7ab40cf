"""The reference's default sampler, `sampler="importance"` (src/aspire/samplers/importance.py:6-22): draw from the
proposal flow, evaluate prior and likelihood, attach importance weights (src/aspire/samples.py:457-475).

The reductions behind `Samples.compute_weights` run in the same HIP kernels as the SMC path (one tempering step from
beta 0 to 1): log_w = ll + lp - log_q, log Z = logsumexp(log_w) - log N, the evidence error
sqrt(sum (w - Z)^2 / (N (N - 1))) from the centred second moment, ESS = exp(2 LSE - LSE(2 .)).

Status: OUTSIDE the accelerated hot path (SURVEY.md §2 lists the reference's importance sampler as out of scope).  It is
kept because the facade's default is `sampler="importance"` and the bounded-prior tests draw through it; it is not
benchmarked and carries no parity claim beyond those tests.
"""
from __future__ import annotations

import math

import numpy as np

from .. import smc_math
from ..samples import Samples
from .base import Sampler, track_calls


class ImportanceSampler(Sampler):
    @track_calls
    def sample(self, n_samples: int) -> Samples:
        e, comm = self.engine, self.comm
        x, log_q = self.prior_flow.sample_and_log_prob(n_samples)
        x_dev = e.asarray(x, dtype=self.x_torch_dtype)
        lq = self._to_dev(log_q)
        lp, ll = self._eval_prior_likelihood(x_dev, lq)
        n = x_dev.shape[0] * comm.world
        st = smc_math.global_stats(e, comm, ll, lp, lq, 0.0, [1.0], n)[0]  # lw(beta 0 -> 1) = (ll + lp) - lq
        samples = Samples(x_dev, log_likelihood=ll, log_prior=lp, xp=e_torch(), parameters=self.parameters)
        samples.log_q = lq  # attached afterwards: the weights come from the kernels below, not from array ops
        samples.log_w = e.log_weights(ll, lp, lq, 0.0, 1.0, 0.0)
        samples.weights = e.normalized_weights(ll, lp, lq, 0.0, 1.0, 0.0, 0.0)  # exp(log_w)
        samples.log_evidence = smc_math.log_evidence_ratio(st)  # logsumexp(log_w) - log N
        samples.evidence = math.exp(samples.log_evidence)
        # sum (w - Z)^2 = e^{2m} sum (e^{lw - m} - Z e^{-m})^2  with Z e^{-m} = S1 / N
        m2 = e.weights_m2(ll, lp, lq, 0.0, 1.0, st.m, st.S1 / n)
        if comm.sharded:
            m2 = float(np.sum(comm.all_gather_f64(np.array([m2]))))
        samples.evidence_error = math.exp(st.m) * math.sqrt(m2 / (n * (n - 1))) if n > 1 else float("nan")
        samples.log_evidence_error = abs(samples.evidence_error / samples.evidence) if samples.evidence else float("nan")
        samples.effective_sample_size = math.exp(2.0 * math.log(st.S1) - math.log(st.S2))
        return samples


def e_torch():
    import torch

    return torch
