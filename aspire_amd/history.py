"""Sampler diagnostics containers (reference src/aspire/history.py:13-81).

The fields the SMC loop appends to (history.py:71-81) and the HDF5 layout of `save` / `load`
(history.py:20-35, 83-149), written through the h5py group protocol (`aspire_amd/io.py`); plotting is
out of scope.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field


@dataclass
class History:
    def save(self, h5_file, path="history"):
        """history.py:20-35: one flattened dataset per field under the group `path`."""
        from .io import recursively_save_to_h5_file

        recursively_save_to_h5_file(h5_file, path, copy.deepcopy(self.__dict__))

    @classmethod
    def load(cls, h5_file, path="history"):
        """history.py:22-48: the dataclass's own fields through its constructor, anything else in the group as plain attributes."""
        from .io import load_from_h5_file

        dictionary = load_from_h5_file(h5_file, path)
        names = set(cls.__dataclass_fields__)
        inst = cls(**{k: v for k, v in dictionary.items() if k in names})
        for k, v in dictionary.items():
            if k not in names:
                setattr(inst, k, v)
        return inst


@dataclass
class FlowHistory(History):
    training_loss: list[float] = field(default_factory=list)
    validation_loss: list[float] = field(default_factory=list)

    def save(self, h5_file, path="flow_history"):
        """history.py:66-68."""
        super().save(h5_file, path=path)


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

    def save(self, h5_file, path="smc_history"):
        """history.py:83-112: every list as a dataset under `path`, the sample history (when kept) as one group per
        iteration under `<path>__sample_history/<i>`, its length under `__len_sample_history`."""
        from .io import recursively_save_to_h5_file

        dictionary = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if k != "sample_history"}
        dictionary["__len_sample_history"] = len(self.sample_history)
        recursively_save_to_h5_file(h5_file, path, dictionary)
        for i, samples in enumerate(self.sample_history):
            samples.save(h5_file, path=f"{path}__sample_history/{i}")

    @classmethod
    def load(cls, h5_file, path="smc_history"):
        """history.py:114-149."""
        from .io import load_from_h5_file
        from .samples import SMCSamples

        dictionary = load_from_h5_file(h5_file, path)
        n_samples = int(dictionary.pop("__len_sample_history", 0))
        dictionary["sample_history"] = [SMCSamples.load(h5_file, path=f"{path}__sample_history/{i}") for i in range(n_samples)]
        names = set(cls.__dataclass_fields__)
        conv = lambda v: v.tolist() if hasattr(v, "tolist") else v  # noqa: E731  (datasets come back as arrays)
        inst = cls(**{k: conv(v) for k, v in dictionary.items() if k in names})
        for k, v in dictionary.items():
            if k not in names:
                setattr(inst, k, v)
        return inst
