"""The reference's integration scenarios (tests/test_reference_integration.py: `test_integration.py:11-48`, `test_checkpointing.py`)
on the HIP engine: `Aspire(flow_backend="zuko", prior_bounds=..., bounded_to_unbounded=..., dtype=...)` -> `fit` ->
`sample_posterior(sampler=..., adaptive=True, sampler_kwargs={"n_steps": 10})` with Python callables, numpy and torch samples, then
config / flow / samples into one file; and an interrupted-run-free resume through `auto_checkpoint` / `resume_from_file`."""
import numpy as np
import pytest
import torch

import test_reference_integration as T
from fake_h5 import FakeFile

pytestmark = pytest.mark.gpu


@pytest.fixture
def h5gpu(monkeypatch, hip_engine):
    from aspire_amd import io
    from aspire_amd import samples as samples_mod

    monkeypatch.setattr(io, "open_h5", lambda path, mode="r": FakeFile(path, mode))
    monkeypatch.setattr(io, "h5py_available", lambda: True)
    monkeypatch.setattr(samples_mod, "_default_engine", hip_engine)
    return io


@pytest.mark.parametrize("sampler", ["smc", "minipcn_smc"])
@pytest.mark.parametrize("xp_name", ["numpy", "torch"])
@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
@pytest.mark.parametrize("dtype", [None, "float32", "float64"])
def test_integration_zuko_on_the_hip_engine(h5gpu, hip_engine, tmp_path, dtype, bounded_to_unbounded, xp_name, sampler):
    from aspire_amd import Aspire, Samples

    dims, parameters, prior_bounds, xp, log_likelihood, log_prior, init = T._fixtures(xp_name)
    samples = Samples(init if xp is np else torch.as_tensor(init), xp=xp)
    aspire = Aspire(log_likelihood=log_likelihood, log_prior=log_prior, dims=dims, parameters=parameters, prior_bounds=prior_bounds,
                    flow_matching=False, bounded_to_unbounded=bounded_to_unbounded, flow_backend="zuko", dtype=dtype)
    aspire.fit(samples, n_epochs=5)
    out = aspire.sample_posterior(n_samples=100, sampler=sampler, adaptive=True, sampler_kwargs={"n_steps": 10}, engine=hip_engine,
                                  rng=np.random.default_rng(3))
    assert len(out.x) == 100 and out.parameters == parameters and np.isfinite(float(out.log_evidence))
    x = np.asarray(out.x if xp is np else out.x.cpu())
    assert np.all(np.abs(x) <= 10.0) and abs(x.mean() - 2.0) < 0.6
    with h5gpu.open_h5(tmp_path / "test_integration_zuko.h5", "w") as h5_file:
        aspire.save_config(h5_file)
        aspire.save_flow(h5_file)
        samples.save(h5_file, path="posterior_samples")
        assert {"aspire_config", "flow", "posterior_samples"} <= set(h5_file.keys())


@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
def test_resume_from_file_smc_on_the_hip_engine(h5gpu, hip_engine, tmp_path, bounded_to_unbounded):
    """test_checkpointing.py:4-46 with the HIP engine under the sampler."""
    from aspire_amd import Aspire

    aspire, kw, samples, checkpoint_file = T._writer(h5gpu, tmp_path, "ckpt.h5", bounded_to_unbounded)
    with aspire.auto_checkpoint(checkpoint_file, every=1):
        aspire.sample_posterior(n_samples=20, sampler="smc", n_final_samples=25, sampler_kwargs={"n_steps": 10, "step_fn": "pcn"},
                                engine=hip_engine, rng=np.random.default_rng(1))
    resumed = Aspire.resume_from_file(checkpoint_file, log_likelihood=kw["log_likelihood"], log_prior=kw["log_prior"])
    with resumed.auto_checkpoint(checkpoint_file, every=1):
        resumed_samples = resumed.sample_posterior(sampler="smc", engine=hip_engine, rng=np.random.default_rng(2))
    assert len(resumed_samples.x) == 25
