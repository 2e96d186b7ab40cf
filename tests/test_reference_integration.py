"""The reference's own integration scenarios, restated against this repository's facade on the CPU test double.

`/root/reference/tests/integration_tests/test_integration.py:11-48` (`test_integration_zuko`: fit 5 epochs, `sample_posterior(
n_samples=100, sampler=..., adaptive=True, sampler_kwargs={"n_steps": 10})`, then `save_config` / `save_flow` / `samples.save`
into one HDF5 file) over its fixtures (`conftest.py`: 2-D Gaussian likelihood mean 2, uniform prior on [-10, 10]^2,
`bounded_to_unbounded` in {True, False}, `dtype` in {None, float32, float64}, numpy and torch samples), for the two sampler names
of the SMC path (`sampler_config`: "smc", "minipcn_smc"); and `test_checkpointing.py` (five tests around `auto_checkpoint` /
`resume_from_file`, `n_final_samples`).  The test double stands in for the HIP engine (`engine=`), an in-memory stand-in for h5py;
everything else - constructor keywords, keyword routing, defaults, return types - is the reference test's text.
tests/test_likelihood_hole.py holds the third scenario of that file (the likelihood hole).
"""
import math

import numpy as np
import pytest
import torch

from fake_h5 import FakeFile
from oracle_engine import OracleEngine


def _fixtures(xp_name):
    dims, mean, std = 2, 2.0, 1.0
    parameters = [f"x_{i}" for i in range(dims)]
    prior_bounds = {p: [-10, 10] for p in parameters}
    xp = np if xp_name == "numpy" else torch

    def log_likelihood(samples):
        assert samples.x.shape[-1] == dims
        x = xp.asarray(samples.x)
        constant = math.log(1 / (std * math.sqrt(2 * math.pi)))
        return xp.sum(constant - (0.5 * ((x - mean) / std) ** 2), axis=-1) if xp is np else (constant - 0.5 * ((x - mean) / std) ** 2).sum(-1)

    def log_prior(samples):
        assert samples.x.shape[-1] == dims
        x = xp.asarray(samples.x)
        constant = dims * math.log(1 / 10)
        if xp is np:
            return np.sum(np.where((x >= -10) & (x <= 10), constant, -np.inf), axis=-1)
        return torch.where((x >= -10) & (x <= 10), torch.full_like(x, constant), torch.full_like(x, -math.inf)).sum(-1)

    init = np.random.default_rng(42).normal(mean, std, size=(500, dims))
    return dims, parameters, prior_bounds, xp, log_likelihood, log_prior, init


@pytest.fixture
def h5(monkeypatch):
    from aspire_amd import io

    monkeypatch.setattr(io, "open_h5", lambda path, mode="r": FakeFile(path, mode))
    monkeypatch.setattr(io, "h5py_available", lambda: True)
    from aspire_amd import samples as samples_mod

    monkeypatch.setattr(samples_mod, "_default_engine", OracleEngine())  # (what `get_default_engine()` hands out: no GPU in this suite)
    return io


@pytest.mark.parametrize("sampler", ["smc", "minipcn_smc"])
@pytest.mark.parametrize("xp_name", ["numpy", "torch"])
@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
@pytest.mark.parametrize("dtype", [None, "float32", "float64"])
def test_integration_zuko(h5, tmp_path, dtype, bounded_to_unbounded, xp_name, sampler):
    from aspire_amd import Aspire, Samples

    dims, parameters, prior_bounds, xp, log_likelihood, log_prior, init = _fixtures(xp_name)
    samples = Samples(init if xp is np else torch.as_tensor(init), xp=xp)
    aspire = Aspire(log_likelihood=log_likelihood, log_prior=log_prior, dims=dims, parameters=parameters, prior_bounds=prior_bounds,
                    flow_matching=False, bounded_to_unbounded=bounded_to_unbounded, flow_backend="zuko", dtype=dtype)
    aspire.fit(samples, n_epochs=5)
    out = aspire.sample_posterior(n_samples=100, sampler=sampler, adaptive=True, sampler_kwargs={"n_steps": 10},
                                  engine=OracleEngine(), rng=np.random.default_rng(3))  # (engine: the test double; no GPU here)
    assert len(out.x) == 100 and out.parameters == parameters and np.isfinite(float(out.log_evidence))
    x = np.asarray(out.x if xp is np else out.x.cpu())
    assert np.all(np.abs(x) <= 10.0) and abs(x.mean() - 2.0) < 0.6
    with h5.open_h5(tmp_path / "test_integration_zuko.h5", "w") as h5_file:
        aspire.save_config(h5_file)
        aspire.save_flow(h5_file)
        samples.save(h5_file, path="posterior_samples")
        assert {"aspire_config", "flow", "posterior_samples"} <= set(h5_file.keys())


def _writer(h5, tmp_path, name, bounded_to_unbounded, fit_to_file=False, **sample_kw):
    from aspire_amd import Aspire, Samples

    dims, parameters, prior_bounds, xp, log_likelihood, log_prior, init = _fixtures("numpy")
    kw = dict(log_likelihood=log_likelihood, log_prior=log_prior, dims=2, parameters=parameters, prior_bounds=prior_bounds,
              bounded_to_unbounded=bounded_to_unbounded, flow_backend="zuko")
    samples = Samples(init, xp=np)
    checkpoint_file = tmp_path / name
    open(checkpoint_file, "wb").close()  # (the in-memory stand-in has no file on disk; the resume code asks the file system)
    aspire = Aspire(**kw)
    if fit_to_file:
        aspire.fit(samples, checkpoint_path=checkpoint_file, n_epochs=10)
    else:
        aspire.fit(samples, n_epochs=10)
    return aspire, kw, samples, checkpoint_file


def _first_run(aspire, checkpoint_file):
    with aspire.auto_checkpoint(checkpoint_file, every=1):
        return aspire.sample_posterior(n_samples=20, sampler="smc", n_final_samples=25, sampler_kwargs={"n_steps": 10, "step_fn": "pcn"},
                                       engine=OracleEngine(), rng=np.random.default_rng(1))


@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
def test_resume_from_file_smc(h5, tmp_path, bounded_to_unbounded):
    """test_checkpointing.py:4-46."""
    from aspire_amd import Aspire

    aspire, kw, samples, checkpoint_file = _writer(h5, tmp_path, "ckpt.h5", bounded_to_unbounded)
    _first_run(aspire, checkpoint_file)
    resumed = Aspire.resume_from_file(checkpoint_file, log_likelihood=kw["log_likelihood"], log_prior=kw["log_prior"])
    with resumed.auto_checkpoint(checkpoint_file, every=1):
        resumed_samples = resumed.sample_posterior(sampler="smc", engine=OracleEngine(), rng=np.random.default_rng(2))
    assert len(resumed_samples.x) == 25


@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
def test_resume_from_file_manual_call(h5, tmp_path, bounded_to_unbounded):
    """test_checkpointing.py:49-90: no checkpoint arguments on the resumed call - the defaults are primed."""
    from aspire_amd import Aspire

    aspire, kw, samples, checkpoint_file = _writer(h5, tmp_path, "ckpt_manual.h5", bounded_to_unbounded)
    _first_run(aspire, checkpoint_file)
    resumed = Aspire.resume_from_file(checkpoint_file, log_likelihood=kw["log_likelihood"], log_prior=kw["log_prior"])
    resumed_samples = resumed.sample_posterior(sampler="smc", engine=OracleEngine(), rng=np.random.default_rng(2))
    assert len(resumed_samples.x) == 25


@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
def test_auto_checkpoint_resume_same_instance(h5, tmp_path, bounded_to_unbounded):
    """test_checkpointing.py:93-126."""
    aspire, kw, samples, checkpoint_file = _writer(h5, tmp_path, "ckpt_same_instance.h5", bounded_to_unbounded)
    _first_run(aspire, checkpoint_file)
    assert not hasattr(aspire, "_resume_from_default")
    with aspire.auto_checkpoint(checkpoint_file, every=1, resume=True):
        assert aspire._resume_n_samples == 20
        resumed_samples = aspire.sample_posterior(sampler="smc", engine=OracleEngine(), rng=np.random.default_rng(2))
    assert len(resumed_samples.x) == 25
    assert not hasattr(aspire, "_resume_from_default")


@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
def test_auto_checkpoint_resume_loads_flow_for_new_instance(h5, tmp_path, bounded_to_unbounded):
    """test_checkpointing.py:129-171."""
    from aspire_amd import Aspire

    writer, kw, samples, checkpoint_file = _writer(h5, tmp_path, "ckpt_new_instance.h5", bounded_to_unbounded, fit_to_file=True)
    _first_run(writer, checkpoint_file)
    resumed = Aspire(**kw)
    assert resumed.flow is None
    with resumed.auto_checkpoint(checkpoint_file, every=1, resume=True):
        assert resumed.flow is not None
        resumed_samples = resumed.sample_posterior(sampler="smc", engine=OracleEngine(), rng=np.random.default_rng(2))
    assert len(resumed_samples.x) == 25


@pytest.mark.parametrize("bounded_to_unbounded", [True, False])
def test_auto_checkpoint_resume_skips_flow_training(h5, tmp_path, bounded_to_unbounded):
    """test_checkpointing.py:174-215."""
    from aspire_amd import Aspire

    writer, kw, samples, checkpoint_file = _writer(h5, tmp_path, "ckpt_skip_fit.h5", bounded_to_unbounded, fit_to_file=True)
    resumed = Aspire(**kw)
    with resumed.auto_checkpoint(checkpoint_file, resume=True):
        original_fit = resumed.flow.fit

        def fail_fit(*args, **kwargs):
            raise AssertionError("flow.fit should not be called")

        resumed.flow.fit = fail_fit
        history = resumed.fit(samples, n_epochs=10)
        resumed.flow.fit = original_fit
        assert history.training_loss == []
        assert history.validation_loss == []


# ---- /root/reference/tests/test_history.py:10-41 (the save / load tests; plotting is out of scope) ------------------------------
def test_history_save_load(h5, tmp_path):
    from aspire_amd.history import History

    history = History()
    history.stat = [1, 2, 3]
    with h5.open_h5(tmp_path / "history.h5", "w") as f:
        history.save(f, path="history")
    with h5.open_h5(tmp_path / "history.h5", "r") as f:
        loaded_history = History.load(f, path="history")
    assert np.array_equal(loaded_history.stat, [1, 2, 3])


def test_smc_history_save_load(h5, tmp_path):
    from aspire_amd.history import SMCHistory
    from aspire_amd.samples import SMCSamples

    history = SMCHistory()
    samples = SMCSamples(x=np.array([[1, 2], [3, 4]]), beta=0.5, parameters=["x1", "x2"])
    history.sample_history.append(samples)
    with h5.open_h5(tmp_path / "smc_history.h5", "w") as f:
        history.save(f, path="smc_history")
    with h5.open_h5(tmp_path / "smc_history.h5", "r") as f:
        loaded_history = SMCHistory.load(f, path="smc_history")
    assert isinstance(loaded_history.sample_history[0], SMCSamples)
    assert len(loaded_history.sample_history) == 1
    assert np.array_equal(loaded_history.sample_history[0].x, [[1, 2], [3, 4]])
    assert loaded_history.sample_history[0].beta == 0.5
    assert loaded_history.sample_history[0].parameters == ["x1", "x2"]


# ---- /root/reference/tests/test_flows/test_torch_flows/test_zuko_flows.py (the flow class the reference builds by default) -------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_zuko_flow(h5, dtype):
    from aspire_amd.flows import MAFFlow
    from aspire_amd.transforms import FlowTransform

    dims = 3
    parameters = [f"x_{i}" for i in range(dims)]
    data_transform = FlowTransform(parameters=parameters, xp=torch, dtype=dtype)
    flow = MAFFlow(dims=dims, seed=42, device="cpu", data_transform=data_transform)
    x = torch.randn(100, dims, device=flow.device)
    flow.fit_data_transform(x)
    assert flow.dims == dims
    x = torch.tensor([0.1, 0.2, 0.3], device=flow.device)
    log_prob = flow.log_prob(x)
    assert log_prob.shape == (1,)


def test_zuko_flow_save_and_load(h5, tmp_path):
    from aspire_amd.flows import MAFFlow

    flow = MAFFlow(dims=2, seed=42, device="cpu")
    x = torch.randn(100, 2, device=flow.device)
    with h5.open_h5(tmp_path / "result.h5", "w") as f:
        flow.save(f, "flow")
    with h5.open_h5(tmp_path / "result.h5", "r") as f:
        loaded_flow = MAFFlow.load(f, "flow")
    assert loaded_flow.dims == flow.dims
    assert torch.allclose(torch.as_tensor(flow.log_prob(x)), torch.as_tensor(loaded_flow.log_prob(x)))
