"""Two ranks sharing ONE MI355X (gloo-staged collectives): the sharded path through the real HIP kernels, checked
against the same two-rank run on the CPU oracle test double (tests/test_dist_gloo.py).  The production backend is
"nccl" (RCCL); one GPU cannot host two RCCL ranks, so the collectives are staged through the host here - the kernels,
the record layouts and the host logic are the ones a multi-GPU run uses."""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

from test_dist_gloo import _worker, spawn_ranks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = {}
    for kind in ("oracle", "hip"):
        d = str(tmp_path_factory.mktemp("dist_" + kind))
        spawn_ranks(_worker, 2, lambda port, d=d, kind=kind: (2, port, d, kind))
        out[kind] = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(2)]
    return out


def test_sharded_search_and_owner_resampling_match_oracle(runs):
    for r in range(2):
        h, o = runs["hip"][r], runs["oracle"][r]
        assert float(h["fb"][0]) == float(o["fb"][0])  # same bisection decisions on the merged rank records
        assert float(h["fb"][1]) == pytest.approx(float(o["fb"][1]), rel=1e-11)
        assert np.array_equal(h["own_idx"], o["own_idx"])  # tile records -> replicated chain -> slice; ordered select; search
        assert np.array_equal(h["own_idx_repl"], o["own_idx"])  # forced fallback (replicated scan): the same ancestors
        assert np.array_equal(h["own_idx_sys"], o["own_idx_sys"])
        assert np.array_equal(h["own_x"], o["own_x"])
        assert h["own_counts"].tolist() == o["own_counts"].tolist() and h["own_gid0"].tolist() == o["own_gid0"].tolist()
        assert np.array_equal(h["again_x"], o["again_x"]) and h["again_counts"].tolist() == o["again_counts"].tolist()
        assert np.array_equal(h["slots_x"], o["slots_x"])
        assert float(h["own_var"]) == pytest.approx(float(o["own_var"]), rel=1e-10)
        assert h["own_flags"].tolist() == o["own_flags"].tolist()
        assert np.array_equal(h["reb_x"], o["reb_x"])
        assert h["again_n"].tolist() == o["again_n"].tolist() and h["skew"].tolist() == o["skew"].tolist()
        assert np.array_equal(h["idx_exact"], o["idx_exact"])  # slot layout


def test_sharded_sampler_matches_oracle(runs):
    for r in range(2):
        h, o = runs["hip"][r], runs["oracle"][r]
        for key in ("beta", "own_beta"):
            np.testing.assert_allclose(h[key], o[key], rtol=1e-8)
        assert float(h["logz"]) == pytest.approx(float(o["logz"]), abs=1e-7)
        assert float(h["own_logz"]) == pytest.approx(float(o["own_logz"]), abs=1e-7)
        assert int(h["own_n"]) == int(o["own_n"])


def test_sharded_step_as_one_chain_equals_the_phase_by_phase_path(runs):
    """smc_math.shard_step_enqueue / shard_step_wait / shard_step_check (search -> moments -> weights -> cdf slice -> draw selection with the
    scalars left on the device, one synchronisation) against find_beta_sharded + resample_owner (a host decision after every
    phase): bit-identical beta*, evidence variance, ancestors and generator state on both ranks.  Contract:
    /root/reference/src/aspire/samples.py:1221-1287 on the global population."""
    for r in range(2):
        h = runs["hip"][r]
        assert h["chain_flags"].tolist() == [1, 1, 1]
        assert h["chain_beta"][0] == h["chain_beta"][1] == float(h["fb"][0])
        assert h["chain_var"][0] == h["chain_var"][1]
        assert np.array_equal(h["chain_x"], h["steps_x"]) and np.array_equal(h["chain_ll"], h["steps_ll"])
        assert h["chain_counts"].tolist() == h["steps_counts"].tolist()
        assert h["chain_rng"][0] == h["chain_rng"][1]


def test_sharded_chain_hands_a_step_it_cannot_take_back_to_the_phase_by_phase_path(runs):
    """Rank 1 holds all the weight (log-likelihood + 400): the search needs more rounds than the chain enqueues and the shares
    leave 1/world (1 +- 25 %).  The chain still runs to its end on the device (uniform weights behind an unfinished search: every
    later launch stays well defined, nothing hangs), reports the step as not taken and leaves the generator alone; the search and
    resample() then go phase by phase exactly as without the chain."""
    for r in range(2):
        h = runs["hip"][r]
        assert h["chain_skew_flags"].tolist() == [1, 0], h["chain_skew_flags"]
        assert h["chain_skew_beta"][0] == h["chain_skew_beta"][1]
        assert np.array_equal(h["chain_skew_x"], h["steps_skew_x"])
        assert h["chain_skew_rng"][0] == h["chain_skew_rng"][1]


def test_sharded_flow_sampler_chain_form_equals_phase_by_phase_form(runs):
    """Two ranks, coupling-flow proposal, the one-kernel flow step with the accept counts exchanged through the callback hook:
    the sampler with the importance step as one chain behind the mutation (and the reference factorisation behind its moments)
    against ASMC_SHARD_STEP=0, where nothing runs ahead - bit-identical schedule, acceptance history, log Z and particles."""
    for r in range(2):
        h = runs["hip"][r]
        assert h["flow_chain_path"].tolist() == [1, 1]
        assert np.array_equal(h["flow_chain_beta"], h["flow_steps_beta"]) and len(h["flow_chain_beta"]) >= 3
        assert np.array_equal(h["flow_chain_acc"], h["flow_steps_acc"])
        assert h["flow_chain_logz"][0] == h["flow_chain_logz"][1]
        assert np.array_equal(h["flow_chain_x"], h["flow_steps_x"])


def test_lagged_adaptation_sharded_matches_the_test_double_and_one_rank(runs):
    """`adapt_lag` (asmc_pcn_params.adapt = k >= 2) on the GPU: (i) the two-rank sampler run with built-in densities
    (k_pcn_adapt's lagged form behind the exchange hook, one exchange per block of k steps) against the same run on the CPU test
    double; (ii) the one-kernel flow step of a sharded run (rank counts in cell t % k, the block's replay in the next step's
    prologue, k_pcn_adapt closing the ragged tail) against ONE rank stepping the whole population with the same lag: the same
    global accept counts, the same step-size history to the bit, the same rows."""
    from conftest import random_coupling_flow

    from aspire_amd.engine import HipEngine

    for r in range(2):
        h, o = runs["hip"][r], runs["oracle"][r]
        np.testing.assert_allclose(h["lag_beta"], o["lag_beta"], rtol=1e-8)
        np.testing.assert_allclose(h["lag_rho"], o["lag_rho"], rtol=1e-9)
        np.testing.assert_allclose(h["lag_acc"], o["lag_acc"], atol=2e-3)
        assert float(h["lag_logz"]) == pytest.approx(float(o["lag_logz"]), abs=1e-7)
    eng = HipEngine(0, n_max=8192, d_max=32)
    dl, nl = 8, 6000
    fl = random_coupling_flow(dl, 2, 64, seed=9)
    dev = fl.device_coupling(eng)
    x = eng.asarray(0.9 * np.random.default_rng(77).normal(size=(nl, dl)))
    t = eng.make_mixture([0.0], np.zeros((1, dl)), np.ones((1, dl)))
    ll, lp, lq = eng.mixture_logpdf(x, t), eng.mixture_logpdf(x, t), eng.coupling_logprob(x, dev)
    eye = eng.asarray(np.eye(dl))
    acc, hist, rho = eng.pcn_mutate_flow(x, ll, lp, lq, 0.4, eng.asarray(np.zeros(dl)), eye, eye, t, t, dev, 31, 0, 0.6, 8, 3, 0.9, 3, "f64", 0.0)
    h0, h1 = runs["hip"]
    for h in (h0, h1):
        assert np.array_equal(h["lagflow_acc"], np.asarray(acc)) and np.array_equal(h["lagflow_hist"], np.asarray(hist))
        assert float(h["lagflow_rho"][0]) == rho
    assert np.array_equal(np.concatenate([h0["lagflow_x"], h1["lagflow_x"]]), x.cpu().numpy())
    # the history is the lagged schedule: held for blocks of three steps, the block's updates applied in order at its end
    hist = np.asarray(hist)
    assert hist[0] == hist[1] == hist[2] == 0.6 and hist[3] == hist[4] == hist[5] != 0.6 and hist[6] == hist[7] != hist[5]


def test_flow16_step_sharded_equals_one_rank(runs):
    """The one-kernel flow step above 32 dimensions (k_tpcn_flow16; d = 48 padded to 64, autoregressive proposal) on two ranks with
    the accept counts exchanged through the hook - per step, and per block of a lagged adaptation - against ONE rank stepping the
    whole population: the same global counts, step-size history and rows, bit for bit."""
    from conftest import random_maf_flow

    from aspire_amd.engine import HipEngine

    eng = HipEngine(0, n_max=8192, d_max=64)
    d6, n6 = 48, 4000
    f6 = random_maf_flow(d6, 2, 64, seed=13)
    dev6 = f6.device_coupling(eng)
    x6 = 0.9 * np.random.default_rng(78).normal(size=(n6, d6))
    t6 = eng.make_mixture([0.0], np.zeros((1, d6)), np.ones((1, d6)))
    eye6, mu6 = eng.asarray(np.eye(d6)), eng.asarray(np.zeros(d6))
    h0, h1 = runs["hip"]
    for lag in (1, 3):
        x = eng.asarray(x6)
        ll, lp, lq = eng.mixture_logpdf(x, t6), eng.mixture_logpdf(x, t6), eng.coupling_logprob(x, dev6)
        acc, hist, rho = eng.pcn_mutate_flow(x, ll, lp, lq, 0.4, mu6, eye6, eye6, t6, t6, dev6, 41, 0, 0.5, 7, 2, 0.9, lag, "f64", 5.0)
        for h in (h0, h1):
            assert np.array_equal(h[f"f16_acc_{lag}"], np.asarray(acc)) and np.array_equal(h[f"f16_hist_{lag}"], np.asarray(hist))
            assert float(h[f"f16_rho_{lag}"][0]) == rho
        assert np.array_equal(np.concatenate([h0[f"f16_x_{lag}"], h1[f"f16_x_{lag}"]]), x.cpu().numpy())
