"""Sampler diagnostics containers (reference src/aspire/history.py:13-81).

Only the fields the SMC loop appends to are kept (history.py:71-81); plotting and HDF5 I/O are
out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class History:
    pass


@dataclass
class FlowHistory(History):
    training_loss: list[float] = field(default_factory=list)
    validation_loss: list[float] = field(default_factory=list)


@dataclass
class SMCHistory(History):
    log_norm_ratio: list[float] = field(default_factory=list)
    log_norm_ratio_var: list[float] = field(default_factory=list)
    beta: list[float] = field(default_factory=list)
    ess: list[float] = field(default_factory=list)
    ess_target: list[float] = field(default_factory=list)
    eff_target: list[float] = field(default_factory=list)
    mcmc_autocorr: list[float] = field(default_factory=list)
    mcmc_acceptance: list[float] = field(default_factory=list)
    sample_history: list = field(default_factory=list)
    # additions of this implementation (not in the reference)
    mcmc_step_size: list[float] = field(default_factory=list)
    mcmc_nu: list[float] = field(default_factory=list)  # degrees of freedom of the tpCN reference (inf: Gaussian)
