"""BASELINE configs[3] and configs[4] in their SHARDED form, at size, on the one GPU this pool has: two ranks (gloo-staged
collectives, see tests/test_gpu_dist.py) each own half of the population on cuda:0 and run the sharded path through the real
HIP kernels -

  * configs[3]: 2 x 4M particles, d = 32 - sharded beta search, owner-layout resampling and the row gather against the CPU
    restatement of the reference on the whole 8M population (the same checks as the single-rank
    test_config4_population_8m_is_step_bit_exact: beta* the same float, ancestors = Generator.choice's, rows bit for bit);
  * configs[4]: 2 x 512k particles, d = 128, two-component mixture likelihood - the whole sampler sharded in the slot
    layout (which reproduces a single-rank run's particle order: schedule and log Z must equal the single-rank run's) and
    in the default owner layout (log Z against the closed form);
  * eight ranks on the same GPU, small shards, a uniform-weight population whose cumulative sum sits on the binade
    boundaries of the exact cdf (tile records that fail verification on several ranks): owner layout = Generator.choice.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dist_gloo import spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _setup(rank, world, port, n_max, d_max):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aspire_amd.comm import TorchDistComm
    from aspire_amd.engine import HipEngine

    eng = HipEngine(0, n_max=n_max, d_max=d_max)
    return eng, TorchDistComm(eng.device)


def _worker_config4(rank, world, port, out_dir):
    eng, comm = _setup(rank, world, port, 8_000_000, 32)
    from test_gpu_fullsize import _device_batch

    from aspire_amd import smc_math
    from aspire_amd.samples import SMCSamples

    n, d = 8_000_000, 32
    n_loc = n // world
    lo = rank * n_loc
    # this rank's rows of the SAME 8M batch the single-rank test draws (the draw is keyed by the global particle index)
    x, ll, lp, lq = _device_batch(eng, n, d, seed=5)
    xl, lll, lpl, lql = (t[lo:lo + n_loc].clone() for t in (x, ll, lp, lq))
    full = [t.cpu().numpy() for t in (ll, lp, lq)] if rank == 0 else None
    del x, ll, lp, lq
    torch.cuda.empty_cache()
    res = {}
    b, eff1, conv, passes, n_nan, trip, trip_one = smc_math.find_beta_sharded(eng, comm, lll, lpl, lql, 0.0, 0.5, 1e-6, n)
    res["fb"] = np.array([b, float(conv), float(n_nan)])
    st = smc_math.Stats(*trip, n)
    res["ess"], res["ratio"] = smc_math.ess(st), smc_math.log_evidence_ratio(st)
    pop = SMCSamples(x=xl, log_likelihood=lll, log_prior=lpl, log_q=lql, beta=0.0, xp=torch, engine=eng, comm=comm)
    new, var = pop.resample(b, rng=np.random.default_rng(8), want_variance=True)
    res["own_n"] = np.array([len(new.x), new.n_global, int(new.ragged)])
    res["own_counts"] = np.array(new.shard_counts)
    # the rows themselves would be 1 GB per rank on disk: their global ancestor ids are recovered from the gathered log q
    # (distinct per particle) and compared row by row on the device instead
    idx_own, _, _, cnt = smc_math.resample_owner(eng, comm, lll, lpl, lql, 0.0, b, n, np.random.default_rng(8), st=st)
    res["own_idx"] = eng.to_numpy(idx_own) + lo
    ok = torch.equal(new.x, xl[idx_own]) and torch.equal(new.log_q, lql[idx_own]) and torch.equal(new.log_likelihood, lll[idx_own])
    res["rows_ok"], res["var"] = np.array([int(ok)]), var
    if rank == 0:
        np.savez(os.path.join(out_dir, "full.npz"), ll=full[0], lp=full[1], lq=full[2])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_config4_two_ranks_x_4m_sharded_is_step_bit_exact(oracle, tmp_path):
    out = str(tmp_path)
    spawn_ranks(_worker_config4, 2, lambda port: (2, port, out))
    rs = [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(2)]
    full = np.load(os.path.join(out, "full.npz"))
    ll, lp, lq = full["ll"], full["lp"], full["lq"]
    n = ll.size
    ref_b = oracle.determine_beta(ll, lp, lq, 0.0, beta_tolerance=1e-6, target_efficiency=0.5)
    ref = oracle.resample_indices(ll, lp, lq, 0.0, ref_b.beta, np.random.default_rng(8).random(n))
    for r, res in enumerate(rs):
        assert float(res["fb"][0]) == ref_b.beta and res["fb"][1] == 1 and res["fb"][2] == 0  # the same float as one rank finds
        assert float(res["ess"]) == pytest.approx(oracle.ess_at_beta(ll, lp, lq, 0.0, ref_b.beta), rel=1e-9)
        assert float(res["ratio"]) == pytest.approx(oracle.log_evidence_ratio(ll, lp, lq, 0.0, ref_b.beta), rel=1e-11)
        mine = ref[(ref >= r * n // 2) & (ref < (r + 1) * n // 2)]  # Generator.choice's draws that land in this shard
        assert np.array_equal(res["own_idx"], mine)
        assert res["rows_ok"][0] == 1 and res["own_n"].tolist() == [mine.size, n, 1]
        assert res["own_counts"].tolist() == [int((ref < n // 2).sum()), int((ref >= n // 2).sum())]
        assert float(res["var"]) == pytest.approx(oracle.log_evidence_ratio_variance(ll, lp, lq, 0.0, ref_b.beta), rel=1e-9)


def _worker_config5(rank, world, port, out_dir, n):
    eng, comm = _setup(rank, world, port, n, 128)
    from test_gpu_fullsize import _config5_targets

    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC

    d = 128
    lik, prior, _ = _config5_targets(d)
    res = {}
    for layout in ("slots", "owner"):
        sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=GaussianFlow(d, sigma=3.0, engine=eng, seed=4),
                    xp=np, engine=eng, comm=comm, rng=np.random.default_rng(1))
        sp.shard_layout = layout
        post = sp.sample(n, sampler_kwargs=dict(n_steps=4, step_fn="pcn"), store_sample_history=False)
        res[layout + "_beta"] = np.array(sp.history.beta)
        res[layout + "_logz"] = np.array([float(post.log_evidence), float(post.log_evidence_error)])
        res[layout + "_n"] = np.array([len(post.x)])
        res[layout + "_path"] = np.array([sp.last_mutation_path])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_config5_two_ranks_x_512k_d128_mixture_sampler(hip_engine, tmp_path):
    """1M x 128 with the two-component mixture likelihood, sharded over two ranks: the slot layout reproduces the single-rank
    run (same schedule, log Z to 1e-7 - the reductions are merged in another order); the owner layout selects the same
    ancestors but lays them out by owner, so its noise streams differ: log Z against the closed form."""
    from test_gpu_fullsize import _config5_targets

    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC

    n, d = 1 << 20, 128
    out = str(tmp_path)
    spawn_ranks(_worker_config5, 2, lambda port: (2, port, out, n))
    rs = [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(2)]
    hip_engine.ensure_capacity(n, d)
    lik, prior, true = _config5_targets(d)
    sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=GaussianFlow(d, sigma=3.0, engine=hip_engine, seed=4),
                xp=np, engine=hip_engine, rng=np.random.default_rng(1))
    one = sp.sample(n, sampler_kwargs=dict(n_steps=4, step_fn="pcn"), store_sample_history=False)
    for res in rs:
        assert np.array_equal(res["slots_beta"], rs[0]["slots_beta"]) and np.array_equal(res["owner_beta"], rs[0]["owner_beta"])
        np.testing.assert_allclose(res["slots_beta"], sp.history.beta, rtol=1e-9)
        assert float(res["slots_logz"][0]) == pytest.approx(float(one.log_evidence), abs=1e-7)
        assert res["owner_beta"][-1] == 1.0 and len(res["owner_beta"]) >= 10
        assert abs(float(res["owner_logz"][0]) - true) < 5 * float(res["owner_logz"][1]) + 0.05
    assert int(rs[0]["slots_n"][0]) == int(rs[1]["slots_n"][0]) == n // 2
    assert int(rs[0]["owner_n"][0]) + int(rs[1]["owner_n"][0]) == n


def _worker_eight(rank, world, port, out_dir):
    eng, comm = _setup(rank, world, port, 1 << 16, 8)
    from aspire_amd import smc_math
    from aspire_amd.samples import SMCSamples

    n_loc, d = 4096, 4
    n = world * n_loc
    g = np.random.default_rng(100 + rank)
    x = eng.asarray(g.normal(size=(n_loc, d)))
    res = {"x": eng.to_numpy(x)}
    # (i) a UNIFORM-weight population: every normalised weight is 2^-15 up to rounding, the running sum meets the binade
    # boundaries of the exact cdf at tile edges on ranks 1, 2, 4 and 7
    zero = eng.asarray(np.zeros(n_loc))
    pop = SMCSamples(x=x, log_likelihood=zero, log_prior=zero.clone(), log_q=zero.clone(), beta=0.0, xp=torch, engine=eng, comm=comm)
    eng.profile(True)
    new, _ = pop.resample(0.3, rng=np.random.default_rng(3), want_variance=True)
    res["uni_kernels"] = np.array(sorted(eng.profile_report().keys()))
    eng.profile(False)
    res["uni_x"], res["uni_counts"] = eng.to_numpy(new.x), np.array(new.shard_counts)
    # (ii) heavy-tailed weights over the eight shards
    ll = eng.asarray(3.0 * g.standard_t(3, size=n_loc))
    pop2 = SMCSamples(x=x, log_likelihood=ll, log_prior=zero.clone(), log_q=zero.clone(), beta=0.0, xp=torch, engine=eng, comm=comm)
    new2, _ = pop2.resample(0.25, rng=np.random.default_rng(4), want_variance=True)
    res["ht_ll"], res["ht_x"], res["ht_ragged"] = eng.to_numpy(ll), eng.to_numpy(new2.x), np.array([int(bool(new2.__dict__.get("ragged")))])
    # (iii) the importance step as ONE chain (smc_math.shard_step_enqueue) over eight shards, twice: from equal shards, then from
    # the ragged population the first step leaves (gid offsets, ragged tile lists, the incoming cdf sums of seven lower ranks)
    ll3 = -4.0 * (x * x).sum(1)  # (sharp enough for two temperatures below beta = 1)
    pop3 = SMCSamples(x=x, log_likelihood=ll3, log_prior=zero.clone(), log_q=zero.clone(), beta=0.0, xp=torch, engine=eng, comm=comm)
    rng3 = np.random.default_rng(5)
    took = pop3.speculate_importance_step(0.5, 1e-6, rng3)
    spec = pop3.__dict__.get("_spec") or {}
    b3 = float(spec.get("beta", -1.0))
    pop3.remember_stats(b3, smc_math.Stats(*spec["search"][5], n))
    new3, var3 = pop3.resample(b3, rng=rng3, want_variance=True)
    res["ch_flags"] = np.array([int(bool(took)), int(bool(spec.get("found"))), int(bool(new3.__dict__.get("ragged")))])
    res["ch_beta"], res["ch_ll"], res["ch_x"], res["ch_var"] = np.array([b3]), eng.to_numpy(ll3), eng.to_numpy(new3.x), np.array([var3])
    res["ch_counts"] = np.array(new3.shard_counts)
    took2 = new3.speculate_importance_step(0.5, 1e-6, rng3)
    spec2 = new3.__dict__.get("_spec") or {}
    b4 = float(spec2.get("beta", -1.0))
    new3.remember_stats(b4, smc_math.Stats(*spec2["search"][5], n))
    new4 = new3.resample(b4, rng=rng3)
    res["ch2_flags"] = np.array([int(bool(took2)), int(bool(spec2.get("found")))])
    res["ch2_beta"], res["ch2_ll_in"], res["ch2_x"] = np.array([b4]), eng.to_numpy(new3.log_likelihood), eng.to_numpy(new4.x)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_on_one_gpu_owner_layout_equals_generator_choice(oracle, tmp_path):
    out = str(tmp_path)
    world = 8
    spawn_ranks(_worker_eight, world, lambda port: (world, port, out))
    rs = [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(world)]
    x = np.concatenate([r["x"] for r in rs])
    n = x.shape[0]
    zero = np.zeros(n)
    edges = np.arange(world + 1) * (n // world)
    w = oracle.normalized_weights(zero, zero, zero, 0.0, 0.3)
    ref = np.random.default_rng(3).choice(n, size=n, replace=True, p=w)
    for r, res in enumerate(rs):
        assert np.array_equal(res["uni_x"], x[ref[(ref >= edges[r]) & (ref < edges[r + 1])]])
    ll = np.concatenate([r["ht_ll"] for r in rs])
    w2 = oracle.normalized_weights(ll, zero, zero, 0.0, 0.25)
    ref2 = np.random.default_rng(4).choice(n, size=n, replace=True, p=w2)
    if rs[0]["ht_ragged"][0]:  # owner layout: by owner, in draw order
        for r, res in enumerate(rs):
            assert np.array_equal(res["ht_x"], x[ref2[(ref2 >= edges[r]) & (ref2 < edges[r + 1])]])
    else:  # weight shares outside +-25 %: the slot layout, draw order over equal shards
        assert np.array_equal(np.concatenate([res["ht_x"] for res in rs]), x[ref2])
    # (iii) the one-chain step: beta* identical on every rank, the ancestors Generator.choice picks at that beta, by owner, in
    # draw order; then the same from the ragged population (concatenated rank-major), continuing the same generator
    for res in rs:
        assert res["ch_flags"].tolist() == [1, 1, 1] and res["ch2_flags"].tolist() == [1, 1]
        assert res["ch_beta"][0] == rs[0]["ch_beta"][0] and res["ch2_beta"][0] == rs[0]["ch2_beta"][0]
        assert res["ch_var"][0] == rs[0]["ch_var"][0]
    b3 = float(rs[0]["ch_beta"][0])
    ll3 = np.concatenate([r["ch_ll"] for r in rs])
    g5 = np.random.default_rng(5)
    ref3 = g5.choice(n, size=n, replace=True, p=oracle.normalized_weights(ll3, zero, zero, 0.0, b3))
    for r, res in enumerate(rs):
        assert np.array_equal(res["ch_x"], x[ref3[(ref3 >= edges[r]) & (ref3 < edges[r + 1])]])
        assert res["ch_counts"].tolist() == [int(((ref3 >= edges[q]) & (ref3 < edges[q + 1])).sum()) for q in range(world)]
    assert float(rs[0]["ch_var"][0]) == pytest.approx(oracle.log_evidence_ratio_variance(ll3, zero, zero, 0.0, b3), rel=1e-10)
    x2 = np.concatenate([r["ch_x"] for r in rs])
    ll_in = np.concatenate([r["ch2_ll_in"] for r in rs])
    np.testing.assert_allclose(ll_in, -4.0 * (x2 * x2).sum(1), rtol=1e-13)  # (the gathered values travel with their rows)
    b4 = float(rs[0]["ch2_beta"][0])
    ref4 = g5.choice(n, size=n, replace=True, p=oracle.normalized_weights(ll_in, zero, zero, b3, b4))
    edges2 = np.concatenate([[0], np.cumsum(rs[0]["ch_counts"])])
    for r, res in enumerate(rs):
        assert np.array_equal(res["ch2_x"], x2[ref4[(ref4 >= edges2[r]) & (ref4 < edges2[r + 1])]])
