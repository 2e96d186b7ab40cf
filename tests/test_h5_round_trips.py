"""Round trips of every container the hot path saves (reference utils.py:733-872, samples.py save / load, history.py:20-149,
transforms.py save / load), run twice: on the in-memory stand-in of oracle/fake_h5.py (always) and on REAL h5py where it is
installed (`pytest.importorskip`: absent from the build image, so that leg is skipped there - ADVICE r5 asked for the gated test).
The bodies are the same, so what the real-h5py leg checks is exactly what the stand-in leg is seen to pass.
"""
import numpy as np
import pytest
import torch

from fake_h5 import FakeFile

from aspire_amd import io
from aspire_amd.history import FlowHistory, SMCHistory
from aspire_amd.samples import Samples, SMCSamples
from aspire_amd.transforms import CompositeTransform
from oracle_engine import OracleEngine


def _open_fake(path, mode):
    return FakeFile(str(path), mode)


def _open_real(path, mode):
    h5py = pytest.importorskip("h5py")
    return h5py.File(str(path), mode)


@pytest.fixture(params=["stand-in", "h5py"])
def opener(request):
    if request.param == "h5py":
        pytest.importorskip("h5py")
        return _open_real
    return _open_fake


def test_samples_round_trip(opener, tmp_path):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(50, 3))
    s = SMCSamples(x, parameters=["a", "b", "c"], log_likelihood=rng.normal(size=50), log_prior=rng.normal(size=50), log_q=rng.normal(size=50),
                   beta=0.25, log_evidence=-1.5, log_evidence_error=0.01)
    with opener(tmp_path / "s.h5", "w") as f:
        s.save(f, path="posterior")
    with opener(tmp_path / "s.h5", "r") as f:
        t = SMCSamples.load(f, path="posterior")
    np.testing.assert_array_equal(np.asarray(t.x), x)
    for name in ("log_likelihood", "log_prior", "log_q"):
        np.testing.assert_array_equal(np.asarray(getattr(t, name)), np.asarray(getattr(s, name)))
    assert list(t.parameters) == ["a", "b", "c"] and float(t.beta) == 0.25 and float(t.log_evidence) == -1.5
    # a torch-namespace container comes back in the torch namespace with its dtype
    st = Samples(torch.as_tensor(x, dtype=torch.float32), xp=torch, dtype=torch.float32)
    with opener(tmp_path / "t.h5", "w") as f:
        st.save(f, path="samples")
    with opener(tmp_path / "t.h5", "r") as f:
        tt = Samples.load(f, path="samples")
    assert isinstance(tt.x, torch.Tensor) and tt.x.dtype == torch.float32
    np.testing.assert_array_equal(tt.x.numpy(), x.astype(np.float32))


def test_history_round_trip(opener, tmp_path):
    h = SMCHistory()
    for i in range(4):
        h.beta.append(0.25 * (i + 1)), h.ess.append(100.0 - i), h.log_norm_ratio.append(-0.1 * i), h.log_norm_ratio_var.append(1e-3)
        h.ess_target.append(50.0), h.eff_target.append(0.5), h.mcmc_acceptance.append(0.3), h.mcmc_autocorr.append(1.5)
        h.sample_history.append(SMCSamples(np.full((5, 2), float(i)), beta=0.25 * (i + 1)))
    fh = FlowHistory(training_loss=[3.0, 2.0, 1.5], validation_loss=[3.1, 2.2, 1.9])
    with opener(tmp_path / "h.h5", "w") as f:
        h.save(f)
        fh.save(f)
    with opener(tmp_path / "h.h5", "r") as f:
        g, fg = SMCHistory.load(f), FlowHistory.load(f, path="flow_history")
    assert list(g.beta) == h.beta and list(g.ess) == h.ess and list(g.mcmc_acceptance) == h.mcmc_acceptance
    assert len(g.sample_history) == 4 and float(np.asarray(g.sample_history[2].x)[0, 0]) == 2.0
    assert list(np.asarray(fg.training_loss)) == fh.training_loss and list(np.asarray(fg.validation_loss)) == fh.validation_loss


def test_sampler_state_round_trip(opener, tmp_path):
    state = {"iteration": 3, "beta": 0.4, "x": np.arange(12.0).reshape(4, 3), "rng": np.random.default_rng(5).bit_generator.state,
             "config": {"n_steps": 8, "step_fn": "pcn"}}
    with opener(tmp_path / "c.h5", "w") as f:
        io.dump_state(state, f, path="checkpoint", dsetname="state")
    with opener(tmp_path / "c.h5", "a") as f:  # (the facade appends its groups to the same file afterwards)
        io.recursively_save_to_h5_file(f, "aspire_config", {"dims": 3, "parameters": ["a", "b", "c"], "prior_bounds": None, "eps": 1e-6})
    with opener(tmp_path / "c.h5", "r") as f:
        back = io.load_state(f, "checkpoint", "state")
        cfg = io.load_from_h5_file(f, "aspire_config")
    assert back["iteration"] == 3 and back["config"] == state["config"] and back["rng"] == state["rng"]
    np.testing.assert_array_equal(back["x"], state["x"])
    assert int(cfg["dims"]) == 3 and [str(p) for p in cfg["parameters"]] == ["a", "b", "c"] and cfg["prior_bounds"] is None


@pytest.mark.parametrize("xp,dtype", [(np, np.float64), (torch, torch.float32)])
def test_composite_transform_round_trip(opener, tmp_path, xp, dtype):
    eng = OracleEngine()
    tr = CompositeTransform(parameters=["a", "b", "c"], periodic_parameters=["b"], prior_bounds={"a": [0.0, 2.0], "b": [0.0, 6.0], "c": [-1.0, 1.0]},
                            bounded_to_unbounded=True, affine_transform=True, xp=xp, dtype=dtype, engine=eng)
    x = np.random.default_rng(1).uniform([0.1, 0.1, -0.9], [1.9, 5.9, 0.9], size=(200, 3))
    tr.fit(xp.asarray(x, dtype=dtype))
    with opener(tmp_path / "t.h5", "w") as f:
        tr.save(f, "data_transform")
    with opener(tmp_path / "t.h5", "r") as f:
        back = CompositeTransform.load(f, "data_transform", engine=eng)
    assert back.dtype == tr.dtype and back.parameters == tr.parameters and back.periodic_parameters == tr.periodic_parameters
    y0, j0 = tr.forward(xp.asarray(x, dtype=dtype))
    y1, j1 = back.forward(xp.asarray(x, dtype=dtype))
    np.testing.assert_array_equal(np.asarray(y0), np.asarray(y1))
    np.testing.assert_array_equal(np.asarray(j0), np.asarray(j1))


def test_a_saved_namespace_name_only_resolves_to_an_array_namespace(tmp_path):
    """A file names the array namespace of what it holds (reference utils.py resolve_xp).  Only `numpy`, `torch` and the array_api_compat
    wrappers of the two resolve; any other module name in a file falls back to numpy instead of being imported."""
    from aspire_amd._xp import resolve_xp

    assert resolve_xp("numpy") is np and resolve_xp("torch") is torch
    assert resolve_xp("array_api_compat.torch").__name__.endswith("torch")  # (the wrapper, or torch itself where it is not installed)
    assert resolve_xp("sklearn.externals.array_api_compat.numpy").__name__.endswith("numpy")
    for hostile in ("os", "subprocess", "antigravity", "evil.numpy", "numpy.evil", "torch.hub", "", None, 7):
        assert resolve_xp(hostile) is np
    with FakeFile(str(tmp_path / "n.h5"), "w") as f:
        Samples(np.zeros((3, 2))).save(f, path="s")
        del f["s/xp"]
        f["s"].create_dataset("xp", data="subprocess")  # a tampered file
    with FakeFile(str(tmp_path / "n.h5"), "r") as f:
        assert Samples.load(f, path="s").xp is np
