"""World-size-2 (gloo, CPU) tests of the particle-sharded path: the sharded run must reproduce the
single-rank run — same beta schedule, same global resample indices, same evidence."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import synth
    from oracle_engine import OracleEngine

    from aspire_amd import smc_math
    from aspire_amd.comm import Comm, TorchDistComm
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.samples import gather_global
    from aspire_amd.targets import DiagGaussianMixture

    eng = OracleEngine()
    comm = TorchDistComm(torch.device("cpu"))
    n, d = 4096, 4
    x, ll, lp, lq = synth(n, d, 3)
    lo, hi = rank * n // world, (rank + 1) * n // world
    loc = [eng.asarray(a[lo:hi]) for a in (x, ll, lp, lq)]
    res = {}
    # C1: reductions
    st = smc_math.global_stats(eng, comm, loc[1], loc[2], loc[3], 0.0, [0.1, 1.0], n)
    res["stats"] = np.array([[s.m, s.S1, s.S2] for s in st])
    res["var"] = smc_math.evidence_variance(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.1, st[0])
    # C2/C3: resample, exact and fast
    for mode in ("exact", "fast"):
        idx, j0 = smc_math.resample_indices(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.1, n, np.random.default_rng(5), mode=mode)
        res["idx_" + mode] = idx.numpy()
        res["j0_" + mode] = j0
        if mode == "exact":
            out = gather_global(eng, comm, idx, *loc)
            res["x_out"] = out[0].numpy()
            res["ll_out"] = out[1].numpy()
    # whole sampler, sharded (fused pCN with host-side global adaptation)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                engine=eng, comm=comm, rng=np.random.default_rng(4))
    post = sp.sample(1024, sampler_kwargs=dict(n_steps=3), store_sample_history=False)
    res["beta"] = np.array(sp.history.beta)
    res["logz"] = float(post.log_evidence)
    res["acc"] = np.array(sp.history.mcmc_acceptance)
    res["x_post"] = np.asarray(post.x)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_results(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("gloo"))
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    return [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(2)]


def test_sharded_reductions_match_single_rank(two_rank_results, oracle):
    from conftest import synth
    from oracle_engine import OracleEngine

    from aspire_amd import smc_math
    from aspire_amd.comm import Comm

    eng = OracleEngine()
    x, ll, lp, lq = synth(4096, 4, 3)
    one = smc_math.global_stats(eng, Comm(), *(eng.asarray(a) for a in (ll, lp, lq)), 0.0, [0.1, 1.0], 4096)
    ref = np.array([[s.m, s.S1, s.S2] for s in one])
    for r in two_rank_results:
        assert np.array_equal(r["stats"][:, 0], ref[:, 0])  # global max exact
        np.testing.assert_allclose(r["stats"], ref, rtol=1e-13)
        assert float(r["var"]) == pytest.approx(oracle.log_evidence_ratio_variance(ll, lp, lq, 0.0, 0.1), rel=1e-10)
    assert np.array_equal(two_rank_results[0]["stats"], two_rank_results[1]["stats"])  # bitwise equal on all ranks


def test_sharded_resample_reproduces_global_indices(two_rank_results, oracle):
    from conftest import synth

    x, ll, lp, lq = synth(4096, 4, 3)
    ref = oracle.resample_indices(ll, lp, lq, 0.0, 0.1, np.random.default_rng(5).random(4096))
    got = np.concatenate([two_rank_results[0]["idx_exact"], two_rank_results[1]["idx_exact"]])
    assert np.array_equal(got, ref)  # exact mode: the rank-chained cdf keeps the sequential rounding
    assert int(two_rank_results[1]["j0_exact"]) == 2048
    fast = np.concatenate([two_rank_results[0]["idx_fast"], two_rank_results[1]["idx_fast"]])
    assert (fast != ref).sum() <= 1
    xo = np.concatenate([two_rank_results[0]["x_out"], two_rank_results[1]["x_out"]])
    assert np.array_equal(xo, x[ref])  # all-to-all row exchange delivers the requested rows in order
    assert np.array_equal(np.concatenate([two_rank_results[0]["ll_out"], two_rank_results[1]["ll_out"]]), ll[ref])


def test_sharded_sampler_matches_single_rank(two_rank_results):
    from oracle_engine import OracleEngine

    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    eng = OracleEngine()
    d = 4
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                engine=eng, rng=np.random.default_rng(4))
    post = sp.sample(1024, sampler_kwargs=dict(n_steps=3), store_sample_history=False)
    r0, r1 = two_rank_results
    assert np.array_equal(r0["beta"], r1["beta"])
    np.testing.assert_allclose(r0["beta"], sp.history.beta, rtol=1e-9)
    assert float(r0["logz"]) == pytest.approx(float(post.log_evidence), abs=5e-9)
    np.testing.assert_allclose(r0["acc"], sp.history.mcmc_acceptance, atol=1e-12)
    xs = np.concatenate([r0["x_post"], r1["x_post"]])
    np.testing.assert_allclose(xs, np.asarray(post.x), rtol=1e-9, atol=1e-9)
