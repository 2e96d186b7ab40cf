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


def spawn_ranks(fn, nprocs, make_args, attempts=4):
    """`mp.spawn(fn, args=make_args(port), nprocs)` on a free rendezvous port.  The port is only KNOWN to be free at the moment it is
    picked: another process of the box (or a socket of the previous test still in TIME_WAIT) can take it before rank 0 listens - seen
    once on a GPU box as EADDRINUSE for every test of a module.  A rendezvous that fails that way is repeated on a fresh port."""
    import torch.multiprocessing as mp

    for attempt in range(attempts):
        try:
            return mp.spawn(fn, args=make_args(_free_port()), nprocs=nprocs, join=True)
        except Exception as exc:  # (ProcessRaisedException carries the child's traceback as text)
            if attempt + 1 < attempts and ("EADDRINUSE" in str(exc) or "address already in use" in str(exc).lower()):
                continue
            raise


def _worker(rank, world, port, out_dir, engine_kind="oracle"):
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

    if engine_kind == "hip":  # tests/test_gpu_dist.py: both ranks share cuda:0, collectives staged through gloo
        from aspire_amd.engine import HipEngine

        eng = HipEngine(0, n_max=8192, d_max=32)
        comm = TorchDistComm(eng.device)
    else:
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
        res["idx_" + mode] = eng.to_numpy(idx)
        res["j0_" + mode] = j0
        if mode == "exact":
            out = gather_global(eng, comm, idx, *loc)
            res["x_out"] = eng.to_numpy(out[0])
            res["ll_out"] = eng.to_numpy(out[1])
    # whole sampler, sharded (fused pCN with host-side global adaptation)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                engine=eng, comm=comm, rng=np.random.default_rng(4))
    sp.shard_layout = "slots"  # reproduces the single-rank particle order
    post = sp.sample(1024, sampler_kwargs=dict(n_steps=3), store_sample_history=False)
    res["beta"] = np.array(sp.history.beta)
    res["logz"] = float(post.log_evidence)
    res["acc"] = np.array(sp.history.mcmc_acceptance)
    res["x_post"] = post.x.detach().cpu().numpy() if torch.is_tensor(post.x) else np.asarray(post.x)
    # the same run with a LAGGED step-size adaptation (sampler_kwargs["adapt_lag"] = 3, 7 steps per temperature: two full blocks
    # and a ragged one): one accept-count exchange per block instead of one per step, same bits as the single-rank run with that lag
    if True:  # (both engines: k_pcn_adapt's lagged form behind the exchange hook on the GPU, its restatement on the test double)
        eng.count_exchanges = 0
        spl = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                     engine=eng, comm=comm, rng=np.random.default_rng(4))
        spl.shard_layout = "slots"
        postl = spl.sample(1024, sampler_kwargs=dict(n_steps=7, adapt_lag=3, target_acceptance_rate=0.95), store_sample_history=False)
        res["lag_beta"], res["lag_logz"] = np.array(spl.history.beta), float(postl.log_evidence)
        res["lag_acc"], res["lag_rho"] = np.array(spl.history.mcmc_acceptance), np.array(spl.history.mcmc_step_size)
        res["lag_x"] = postl.x.detach().cpu().numpy() if torch.is_tensor(postl.x) else np.asarray(postl.x)
        res["lag_exchanges"] = np.array([eng.count_exchanges if engine_kind == "oracle" else 3 * len(spl.history.mcmc_acceptance),
                                         len(spl.history.mcmc_acceptance)])
    # ---- owner layout: device-style sharded search, offspring stay on the ancestor's rank ----
    res["fb"] = np.array(smc_math.find_beta_sharded(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.5, 1e-6, n)[:2])
    st01 = smc_math.global_stats(eng, comm, loc[1], loc[2], loc[3], 0.0, [0.1], n)[0]
    idx, var, _, cnt = smc_math.resample_owner(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.1, n, np.random.default_rng(5), st=st01)
    res["own_idx"] = eng.to_numpy(idx) + lo  # global ancestor ids
    res["own_var"] = var
    res["own_counts"] = np.array(cnt)
    # the same through the brute-force path (all-gathered weights, replicated scan): must select the same ancestors
    idx_r, _, _, _ = smc_math.resample_owner(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.1, n, np.random.default_rng(5), st=st01,
                                             force_replicated=True)
    res["own_idx_repl"] = eng.to_numpy(idx_r) + lo
    # opt-in systematic resampling keeps its offspring at home too
    idx_s, _, _, _ = smc_math.resample_owner(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.1, n, np.random.default_rng(5), st=st01,
                                             method="systematic")
    res["own_idx_sys"] = eng.to_numpy(idx_s) + lo
    from aspire_amd.samples import SMCSamples, rebalance_shards

    pop = SMCSamples(x=loc[0], log_likelihood=loc[1], log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
    new, var2 = pop.resample(0.1, rng=np.random.default_rng(5), want_variance=True)
    res["own_x"] = eng.to_numpy(new.x)
    res["own_flags"] = np.array([new.n_global, int(new.ragged), var2 == var])
    res["own_shard_counts"] = np.array(new.shard_counts)
    res["own_gid0"] = np.array([new.gid0()])
    # a different output size (the sampler's n_final_samples): owner layout hands every draw to exactly one rank
    odd = pop.resample(0.1, n_samples=1001, rng=np.random.default_rng(8))
    res["odd_n"] = np.array([len(odd.x), odd.n_global])
    # ragged shards back to equal ones, global order kept
    xb, llb, _, _ = rebalance_shards(eng, comm, new.x, new.log_likelihood, new.log_prior, new.log_q)
    res["reb_x"], res["reb_ll"], res["own_ll"] = eng.to_numpy(xb), eng.to_numpy(llb), eng.to_numpy(new.log_likelihood)
    # a second resampling from the ragged population: owner layout again (ragged in, ragged out) ...
    again = new.resample(0.3, rng=np.random.default_rng(6))
    res["again_x"] = eng.to_numpy(again.x)
    res["again_counts"] = np.array(again.shard_counts)
    # ... and through the slot layout, which hands back equal shards
    slots = new.resample(0.3, rng=np.random.default_rng(6), shard_layout="slots")
    res["again_n"] = np.array([len(slots.x), int(bool(slots.__dict__.get("ragged")))])
    res["slots_x"] = eng.to_numpy(slots.x)
    # uneven weight shares: rank 1 holds almost all the weight -> every rank falls back to the slot layout
    ll_skew = loc[1] + (40.0 if rank == 1 else 0.0)
    pop2 = SMCSamples(x=loc[0], log_likelihood=ll_skew, log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
    sk = pop2.resample(0.5, rng=np.random.default_rng(7))
    res["skew"] = np.array([len(sk.x), int(bool(sk.__dict__.get("ragged")))])
    # every rank arrives with a DIFFERENT generator: sample() hands rank 0's state to everyone
    sp2 = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                 engine=eng, comm=comm, rng=np.random.default_rng(4 + 100 * rank))
    post2 = sp2.sample(1024, sampler_kwargs=dict(n_steps=3), store_sample_history=False)
    res["own_beta"] = np.array(sp2.history.beta)
    res["own_logz"] = float(post2.log_evidence)
    res["own_logz_err"] = float(post2.log_evidence_error)
    res["own_n"] = len(post2.x)
    # coupling-flow proposal trained SEPARATELY (and differently) on every rank: sample() must make the ranks agree on
    # rank 0's parameters before anything is evaluated, and every rank draws its own shard
    if engine_kind == "oracle":
        from aspire_amd.flows import CouplingFlow

        cf = CouplingFlow(d, n_layers=2, hidden_features=(16, 16), seed=11 + rank, dtype=torch.float64)
        cf.fit(1.3 * np.random.default_rng(20 + rank).normal(size=(400, d)), n_epochs=3)
        sp3 = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=cf, xp=np, engine=eng, comm=comm,
                     rng=np.random.default_rng(4))
        post3 = sp3.sample(512, sampler_kwargs=dict(n_steps=2, step_fn="pcn"), store_sample_history=False)
        res["cf_params"] = torch.cat([p.detach().reshape(-1) for p in cf.layers.parameters()]).numpy()
        res["cf_beta"] = np.array(sp3.history.beta)
        res["cf_x0"] = (post3.x.detach().cpu().numpy() if torch.is_tensor(post3.x) else np.asarray(post3.x))[:4]
        # preconditioning="flow" on a sharded population (round-3 advice): ONE fit on a pooled subsample (rank 0 trains, everyone
        # takes its parameters) instead of one per rank thrown away; and building / training a flow leaves the process-wide torch
        # RNG alone (every refit used to call torch.manual_seed: a user density drawing from torch saw repeated streams)
        from aspire_amd.transforms import FlowPreconditioningTransform

        torch.manual_seed(900 + rank)
        probe_before = torch.get_rng_state().clone()
        fpt = FlowPreconditioningTransform(parameters=[f"p{i}" for i in range(d)], flow_backend="coupling", xp=np,
                                           flow_kwargs=dict(n_layers=2, hidden_features=(8, 8)),
                                           fit_kwargs=dict(n_epochs=2, fit_subsample=512), engine=eng)
        z = fpt.fit(np.asarray(x[lo:hi]) + 0.25 * rank, comm=comm)  # the shards differ: a per-rank fit would differ too
        res["fpt_params"] = torch.cat([p.detach().reshape(-1).double() for p in fpt.flow.layers.parameters()]
                                      + [fpt.flow.loc.double().reshape(-1), fpt.flow.scale.double().reshape(-1)]).numpy()
        res["fpt_rng_untouched"] = np.array(int(torch.equal(torch.get_rng_state(), probe_before)))
        res["fpt_z_shape"] = np.array(np.asarray(z).shape)
    if True:  # (both engines: the HIP kernels on the GPU, the host logic of the same chain on the CPU test double)
        # the sharded importance step as ONE chain of launches (smc_math.shard_step_enqueue: the scalars between the phases
        # stay on the device) against the phase-by-phase path above: same beta*, same ancestors, same variance, same generator
        rng_a, rng_b = np.random.default_rng(5), np.random.default_rng(5)
        pop_a = SMCSamples(x=loc[0], log_likelihood=loc[1], log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
        took = pop_a.speculate_importance_step(0.5, 1e-6, rng_a)
        spec = pop_a.__dict__.get("_spec") or {}
        res["chain_flags"] = np.array([int(bool(took)), int(bool(spec.get("found"))), int(spec.get("rows") is not None)])
        b_chain = float(spec.get("beta", -1.0))
        pop_a.remember_stats(b_chain, smc_math.Stats(*spec["search"][5], n))
        new_a, var_a = pop_a.resample(b_chain, rng=rng_a, want_variance=True)
        os.environ["ASMC_SHARD_STEP"] = "0"
        pop_b = SMCSamples(x=loc[0], log_likelihood=loc[1], log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
        assert not pop_b.speculate_importance_step(0.5, 1e-6, rng_b)
        fb = smc_math.find_beta_sharded(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.5, 1e-6, n)
        b_steps = fb[0]
        pop_b.remember_stats(b_steps, smc_math.Stats(*fb[5], n))  # (what the sampler's determine_beta keeps of the search)
        new_b, var_b = pop_b.resample(b_steps, rng=rng_b, want_variance=True)
        del os.environ["ASMC_SHARD_STEP"]
        res["chain_beta"] = np.array([b_chain, b_steps])
        res["chain_var"] = np.array([var_a, var_b])
        res["chain_x"], res["steps_x"] = eng.to_numpy(new_a.x), eng.to_numpy(new_b.x)
        res["chain_ll"], res["steps_ll"] = eng.to_numpy(new_a.log_likelihood), eng.to_numpy(new_b.log_likelihood)
        res["chain_counts"], res["steps_counts"] = np.array(new_a.shard_counts), np.array(new_b.shard_counts)
        res["chain_rng"] = np.array([rng_a.integers(0, 2**62), rng_b.integers(0, 2**62)])
        # a step the chain must hand back: rank 1 holds almost all the weight, so the shares leave 1/world (1 +- 25 %) - the
        # chain runs to its end on the device all the same (nothing hangs, the generator is untouched), reports found = False
        # and resample() takes the phase-by-phase path to the slot layout
        rng_c, rng_d = np.random.default_rng(7), np.random.default_rng(7)
        ll_skew = loc[1] + (400.0 if rank == 1 else 0.0)  # (uneven at ANY beta the search can pick)
        pop_c = SMCSamples(x=loc[0], log_likelihood=ll_skew, log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
        took_c = pop_c.speculate_importance_step(0.5, 1e-6, rng_c)
        spec_c = pop_c.__dict__.get("_spec") or {}
        res["chain_skew_flags"] = np.array([int(bool(took_c)), int(bool(spec_c.get("found")))])
        fb_c = smc_math.find_beta_sharded(eng, comm, ll_skew, loc[2], loc[3], 0.0, 0.5, 1e-6, n)  # (what determine_beta does next)
        pop_c.remember_stats(fb_c[0], smc_math.Stats(*fb_c[5], n))
        sk_c = pop_c.resample(fb_c[0], rng=rng_c)
        os.environ["ASMC_SHARD_STEP"] = "0"
        pop_d = SMCSamples(x=loc[0], log_likelihood=ll_skew, log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
        fb_d = smc_math.find_beta_sharded(eng, comm, ll_skew, loc[2], loc[3], 0.0, 0.5, 1e-6, n)
        pop_d.remember_stats(fb_d[0], smc_math.Stats(*fb_d[5], n))
        sk_d = pop_d.resample(fb_d[0], rng=rng_d)
        del os.environ["ASMC_SHARD_STEP"]
        res["chain_skew_beta"] = np.array([fb_c[0], fb_d[0]])
        res["chain_skew_x"], res["steps_skew_x"] = eng.to_numpy(sk_c.x), eng.to_numpy(sk_d.x)
        res["chain_skew_rng"] = np.array([rng_c.integers(0, 2**62), rng_d.integers(0, 2**62)])
    if engine_kind == "hip":
        # the WHOLE sharded sampler with a coupling-flow proposal (the one-kernel flow step, accept counts exchanged through the
        # Python callback hook): the chain form - enqueued behind the mutation's step loop, factorisation behind its moments -
        # against the phase-by-phase form (ASMC_SHARD_STEP=0: nothing runs ahead): same schedule, same log Z, same particles
        from aspire_amd.flows import CouplingFlow

        def flow_run():
            cf = CouplingFlow(d, n_layers=2, hidden_features=(64, 64), seed=11, dtype=torch.float32, device=eng.device)
            cf.fit(1.3 * np.random.default_rng(20).normal(size=(400, d)), n_epochs=2)
            spf = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=cf, xp=torch, engine=eng, comm=comm,
                         rng=np.random.default_rng(4))
            post = spf.sample(2048, target_efficiency=0.9, sampler_kwargs=dict(n_steps=4, step_fn="pcn"), store_sample_history=False)
            return (np.array(spf.history.beta), float(post.log_evidence), np.array(spf.history.mcmc_acceptance),
                    post.x.detach().cpu().numpy(), spf.last_mutation_path)

        a = flow_run()
        os.environ["ASMC_SHARD_STEP"] = "0"
        b = flow_run()
        del os.environ["ASMC_SHARD_STEP"]
        res["flow_chain_beta"], res["flow_steps_beta"] = a[0], b[0]
        res["flow_chain_logz"] = np.array([a[1], b[1]])
        res["flow_chain_acc"], res["flow_steps_acc"] = a[2], b[2]
        res["flow_chain_x"], res["flow_steps_x"] = a[3], b[3]
        res["flow_chain_path"] = np.array([int("device-side step loop" in a[4]), int("device-side step loop" in b[4])])
    if engine_kind == "hip":
        # lagged adaptation through the ONE-KERNEL flow step of a sharded run (the step's last block leaves the rank's count in
        # cell t % k, the exchange runs at the end of a block, the next block's first step replays the k updates in its
        # prologue): engine level, this rank's half of a fixed population; tests/test_gpu_dist.py compares with one rank
        from conftest import random_coupling_flow

        dl, nl = 8, 6000
        fl = random_coupling_flow(dl, 2, 64, seed=9)
        devl = fl.device_coupling(eng)
        gl = np.random.default_rng(77)
        xl = 0.9 * gl.normal(size=(nl, dl))
        lo2, hi2 = rank * nl // world, (rank + 1) * nl // world
        xs_l = eng.asarray(xl[lo2:hi2])
        tl = eng.make_mixture([0.0], np.zeros((1, dl)), np.ones((1, dl)))
        ll_l, lp_l, lq_l = eng.mixture_logpdf(xs_l, tl), eng.mixture_logpdf(xs_l, tl), eng.coupling_logprob(xs_l, devl)
        eye = eng.asarray(np.eye(dl))
        eng.set_count_hook(comm, nl)
        try:
            acc_l, hist_l, rho_l = eng.pcn_mutate_flow(xs_l, ll_l, lp_l, lq_l, 0.4, eng.asarray(np.zeros(dl)), eye, eye, tl, tl, devl, 31,
                                                      lo2, 0.6, 8, 3, 0.9, 3, "f64", 0.0)
        finally:
            eng.set_count_hook(None, None)
        res["lagflow_acc"], res["lagflow_hist"], res["lagflow_rho"] = np.asarray(acc_l), np.asarray(hist_l), np.array([rho_l])
        res["lagflow_x"] = eng.to_numpy(xs_l)
        # the same through the one-kernel step above 32 dimensions (k_tpcn_flow16: k_count_sum -> exchange -> k_pcn_adapt per
        # step, or per block of a lagged adaptation), d = 48 zero-padded to 64, autoregressive proposal
        from conftest import random_maf_flow

        from aspire_amd.engine import HipEngine as _HE

        eng64 = _HE(0, n_max=8192, d_max=64)
        d6, n6 = 48, 4000
        f6 = random_maf_flow(d6, 2, 64, seed=13)
        dev6 = f6.device_coupling(eng64)
        x6 = 0.9 * np.random.default_rng(78).normal(size=(n6, d6))
        lo6, hi6 = rank * n6 // world, (rank + 1) * n6 // world
        t6 = eng64.make_mixture([0.0], np.zeros((1, d6)), np.ones((1, d6)))
        eye6, mu6 = eng64.asarray(np.eye(d6)), eng64.asarray(np.zeros(d6))
        for lag6 in (1, 3):
            xs6 = eng64.asarray(x6[lo6:hi6])
            l6, p6, q6 = eng64.mixture_logpdf(xs6, t6), eng64.mixture_logpdf(xs6, t6), eng64.coupling_logprob(xs6, dev6)
            eng64.set_count_hook(comm, n6)
            try:
                a6, h6, r6 = eng64.pcn_mutate_flow(xs6, l6, p6, q6, 0.4, mu6, eye6, eye6, t6, t6, dev6, 41, lo6, 0.5, 7, 2, 0.9, lag6, "f64", 5.0)
            finally:
                eng64.set_count_hook(None, None)
            res[f"f16_acc_{lag6}"], res[f"f16_hist_{lag6}"], res[f"f16_rho_{lag6}"] = np.asarray(a6), np.asarray(h6), np.array([r6])
            res[f"f16_x_{lag6}"] = eng64.to_numpy(xs6)
        eng64.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_results(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("gloo"))
    spawn_ranks(_worker, 2, lambda port: (2, port, out))
    return [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(2)]


def test_sharded_step_as_one_chain_equals_the_phase_by_phase_path_on_cpu(two_rank_results):
    """The host logic of the one-chain sharded importance step (smc_math.shard_step_enqueue / _wait / _check,
    SMCSamples.speculate_importance_step / finish_speculation / resample's parked-rows branch) on the CPU test double, two gloo
    ranks: same beta*, evidence variance, ancestors and generator state as find_beta_sharded + resample_owner; a step the chain
    cannot take (all the weight on one rank) is reported as such and redone phase by phase.  The GPU form of the same test runs
    in tests/test_gpu_dist.py.  Contract: /root/reference/src/aspire/samples.py:1221-1287 on the global population."""
    for h in two_rank_results:
        assert h["chain_flags"].tolist() == [1, 1, 1]
        assert h["chain_beta"][0] == h["chain_beta"][1] == float(h["fb"][0])
        assert h["chain_var"][0] == h["chain_var"][1]
        assert np.array_equal(h["chain_x"], h["steps_x"]) and np.array_equal(h["chain_ll"], h["steps_ll"])
        assert h["chain_counts"].tolist() == h["steps_counts"].tolist()
        assert h["chain_rng"][0] == h["chain_rng"][1]
        assert h["chain_skew_flags"].tolist() == [1, 0], h["chain_skew_flags"]
        assert h["chain_skew_beta"][0] == h["chain_skew_beta"][1]
        assert np.array_equal(h["chain_skew_x"], h["steps_skew_x"])
        assert h["chain_skew_rng"][0] == h["chain_skew_rng"][1]


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


def _single_rank_lagged(n, n_steps, lag, seed=4, **sample_kw):
    from oracle_engine import OracleEngine

    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    eng, d = OracleEngine(), 4
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                engine=eng, rng=np.random.default_rng(seed))
    post = sp.sample(n, sampler_kwargs=dict(n_steps=n_steps, adapt_lag=lag, target_acceptance_rate=0.95), store_sample_history=False, **sample_kw)  # (0.95: the step size has to move)
    return sp, post


def test_lagged_adaptation_sharded_equals_single_rank_with_the_same_lag(two_rank_results):
    """`sampler_kwargs["adapt_lag"] = k`: the step size is held for blocks of k steps and the block's k Robbins-Monro updates are
    applied at its end, in order - so a sharded run needs ONE accept-count exchange per block (here 3 per 7-step mutation
    instead of 7).  Two gloo ranks reproduce the single-rank run with the same lag bit for bit (schedule, log Z, acceptance and
    step-size history, particles); the lag changes the adaptation schedule, so it differs from the lag-1 run - which stays the
    default and the specification every other test pins."""
    sp, post = _single_rank_lagged(1024, 7, 3)
    r0, r1 = two_rank_results
    for r in (r0, r1):
        assert np.array_equal(r["lag_beta"], np.array(sp.history.beta)) and float(r["lag_logz"]) == pytest.approx(float(post.log_evidence), abs=5e-9)
        assert np.array_equal(r["lag_acc"], np.array(sp.history.mcmc_acceptance))
        assert np.array_equal(r["lag_rho"], np.array(sp.history.mcmc_step_size))
        assert r["lag_exchanges"][0] == 3 * r["lag_exchanges"][1]  # ceil(7 / 3) exchanges per mutation
    xs = np.concatenate([r0["lag_x"], r1["lag_x"]])
    np.testing.assert_allclose(xs, np.asarray(post.x), rtol=1e-9, atol=1e-9)
    sp1, _ = _single_rank_lagged(1024, 7, 1)
    assert not np.array_equal(np.array(sp1.history.mcmc_step_size), np.array(sp.history.mcmc_step_size))
    with pytest.raises(ValueError, match="adapt_lag"):
        _single_rank_lagged(256, 2, 65)


def test_sharded_beta_search_matches_host_bisection(two_rank_results):
    from conftest import synth
    from oracle_engine import OracleEngine

    from aspire_amd import smc_math
    from aspire_amd.comm import Comm

    eng = OracleEngine()
    x, ll, lp, lq = synth(4096, 4, 3)
    t = [eng.asarray(a) for a in (ll, lp, lq)]

    def eff_fn(betas, closed_form=False):
        return [smc_math.ess(s) / 4096 for s in smc_math.global_stats(eng, Comm(), *t, 0.0, betas, 4096)]

    beta, _, _ = smc_math.determine_beta(eff_fn, 0.0, adaptive=True, beta_step=float("nan"), min_beta_step=0.0,
                                         max_beta_step=1.0, beta_tolerance=1e-6, adaptive_min_beta_step=False, target=0.5,
                                         rate=1.0)
    r0, r1 = two_rank_results
    assert np.array_equal(r0["fb"], r1["fb"])
    assert float(r0["fb"][0]) == beta  # same bisection decisions -> the same float


def test_owner_layout_resampling(two_rank_results, oracle):
    """Default sharded layout: every rank holds EXACTLY the sub-sequence of numpy's Generator.choice index vector that
    points into its shard - same ancestors, same multiplicities, in draw order (samples.py:1277-1278)."""
    from conftest import synth

    n = 4096
    x, ll, lp, lq = synth(n, 4, 3)
    r0, r1 = two_rank_results
    # the reference itself: Generator.choice on the normalised weights of the whole population
    w = oracle.normalized_weights(ll, lp, lq, 0.0, 0.1)
    ref = np.random.default_rng(5).choice(n, size=n, replace=True, p=w)
    assert np.array_equal(ref, oracle.resample_indices(ll, lp, lq, 0.0, 0.1, np.random.default_rng(5).random(n)))
    for r, res in enumerate((r0, r1)):
        mine = ref[(ref >= r * n // 2) & (ref < (r + 1) * n // 2)]  # choice's draws that land in this rank's shard
        assert np.array_equal(res["own_idx"], mine)  # no sort, no tolerance
        assert np.array_equal(res["own_idx_repl"], mine)  # brute-force path (replicated scan): the same selection
        assert np.array_equal(res["own_x"], x[mine])
        assert res["own_counts"].tolist() == [int((ref < n // 2).sum()), int((ref >= n // 2).sum())]
        assert res["own_shard_counts"].tolist() == res["own_counts"].tolist()
    assert int(r0["own_gid0"][0]) == 0 and int(r1["own_gid0"][0]) == int((ref < n // 2).sum())
    # systematic draws through the same owner machinery: the oracle's systematic ancestors, split by owner
    u_sys = oracle.systematic_uniforms(n, float(np.random.default_rng(5).random()))
    ref_sys = oracle.resample_indices(ll, lp, lq, 0.0, 0.1, u_sys)
    assert np.array_equal(np.concatenate([r0["own_idx_sys"], r1["own_idx_sys"]]), ref_sys)  # sorted draws: rank-major order
    assert float(r0["own_var"]) == pytest.approx(oracle.log_evidence_ratio_variance(ll, lp, lq, 0.0, 0.1), rel=1e-10)
    assert int(r0["odd_n"][0]) + int(r1["odd_n"][0]) == 1001 and int(r0["odd_n"][1]) == int(r1["odd_n"][1]) == 1001
    ref_odd = np.random.default_rng(8).choice(n, size=1001, replace=True, p=w)
    assert int(r0["odd_n"][0]) == int((ref_odd < n // 2).sum())
    for res in (r0, r1):
        assert res["own_flags"].tolist() == [n, 1, 1]
        assert res["again_n"].tolist() == [n // 2, 0]  # slot layout hands back equal shards
        assert res["skew"].tolist() == [n // 2, 0]  # imbalance -> slot layout
    # rebalancing keeps the global order and yields equal shards
    allx = np.concatenate([r0["own_x"], r1["own_x"]])
    allll = np.concatenate([r0["own_ll"], r1["own_ll"]])
    assert r0["reb_x"].shape[0] == r1["reb_x"].shape[0] == n // 2
    assert np.array_equal(np.concatenate([r0["reb_x"], r1["reb_x"]]), allx)
    assert np.array_equal(np.concatenate([r0["reb_ll"], r1["reb_ll"]]), allll)
    # second resampling FROM the ragged population: numpy's choice on the concatenated (rank-major) population
    lp2 = -0.5 * np.sum(allx**2, axis=1)
    lq2 = -0.5 * np.sum((allx / 1.5) ** 2, axis=1) - 4 * np.log(1.5) - 0.5 * 4 * np.log(2 * np.pi)
    w2 = oracle.normalized_weights(allll, lp2, lq2, 0.1, 0.3)
    ref2 = np.random.default_rng(6).choice(n, size=n, replace=True, p=w2)
    c0 = int(r0["own_counts"][0])
    assert np.array_equal(r0["again_x"], allx[ref2[ref2 < c0]]) and np.array_equal(r1["again_x"], allx[ref2[ref2 >= c0]])
    assert r0["again_counts"].tolist() == [int((ref2 < c0).sum()), int((ref2 >= c0).sum())]
    # ... and the slot layout selects the same ancestors, laid out in draw order over equal shards
    assert np.array_equal(np.concatenate([r0["slots_x"], r1["slots_x"]]), allx[ref2])


def test_owner_layout_sampler(two_rank_results):
    r0, r1 = two_rank_results
    assert np.array_equal(r0["own_beta"], r1["own_beta"]) and r0["own_beta"][-1] == 1.0
    assert float(r0["own_logz"]) == float(r1["own_logz"])
    assert int(r0["own_n"]) + int(r1["own_n"]) == 1024
    # first temperature: nothing has been permuted yet, the search sees the same population as the slot-layout run
    assert r0["own_beta"][0] == pytest.approx(r0["beta"][0], rel=1e-12)
    analytic = 2.0 * np.log(np.pi)  # (d/2) log pi, d = 4
    assert abs(float(r0["own_logz"]) - analytic) < 5 * float(r0["own_logz_err"]) + 0.05


def test_sharded_run_with_trained_flow_agrees_on_rank0_parameters(two_rank_results):
    r0, r1 = two_rank_results
    assert np.array_equal(r0["cf_params"], r1["cf_params"])  # rank 1 took rank 0's flow
    assert np.array_equal(r0["cf_beta"], r1["cf_beta"]) and r0["cf_beta"][-1] == 1.0
    assert not np.array_equal(r0["cf_x0"], r1["cf_x0"])  # separate draw streams: the shards are not copies of each other


def test_sharded_flow_preconditioning_fits_once_on_a_pooled_subsample(two_rank_results):
    r0, r1 = two_rank_results
    assert np.array_equal(r0["fpt_params"], r1["fpt_params"])  # one latent space for every rank's chain
    assert np.all(np.isfinite(r0["fpt_params"])) and int(r0["fpt_rng_untouched"]) == 1 and int(r1["fpt_rng_untouched"]) == 1
    assert tuple(r0["fpt_z_shape"]) == (2048, 4)


# ---- world 8 (the node's rank count) on the CPU test double --------------------------------------------------------------
def _worker8(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import synth
    from oracle_engine import OracleEngine

    from aspire_amd import smc_math
    from aspire_amd.comm import TorchDistComm
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.samples import SMCSamples
    from aspire_amd.targets import DiagGaussianMixture

    eng, comm = OracleEngine(), TorchDistComm(torch.device("cpu"))
    n, d = world * 4096, 4
    x, ll, lp, lq = synth(n, d, 13)
    lo, hi = rank * n // world, (rank + 1) * n // world
    loc = [eng.asarray(a[lo:hi]) for a in (x, ll, lp, lq)]
    res = {}
    res["fb"] = np.array(smc_math.find_beta_sharded(eng, comm, loc[1], loc[2], loc[3], 0.0, 0.5, 1e-6, n)[:2])
    pop = SMCSamples(x=loc[0], log_likelihood=loc[1], log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
    # owner layout, eight ragged shards
    new, var = pop.resample(0.08, rng=np.random.default_rng(5), want_variance=True)
    res["own_x"], res["own_counts"], res["own_var"] = eng.to_numpy(new.x), np.array(new.shard_counts), var
    res["own_flags"] = np.array([new.n_global, int(new.ragged), new.gid0()])
    # a second resampling FROM the ragged population, to a size the ranks cannot share equally
    odd = new.resample(0.2, n_samples=n - 3, rng=np.random.default_rng(6))
    res["odd_x"], res["odd_counts"] = eng.to_numpy(odd.x), np.array(odd.shard_counts)
    # skew: ranks 2 and 5 hold almost all the weight -> the +-25 % rule sends every rank to the slot layout, which rebalances
    ll_skew = loc[1] + (30.0 if rank in (2, 5) else 0.0)
    pop2 = SMCSamples(x=loc[0], log_likelihood=ll_skew, log_prior=loc[2], log_q=loc[3], beta=0.0, xp=torch, engine=eng, comm=comm)
    sk = pop2.resample(0.5, rng=np.random.default_rng(7))
    res["skew_x"], res["skew_flags"] = eng.to_numpy(sk.x), np.array([len(sk.x), int(bool(sk.__dict__.get("ragged")))])
    # the whole sampler over eight ranks, with a final population the ranks cannot share equally; every rank arrives with
    # its own generator
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                engine=eng, comm=comm, rng=np.random.default_rng(40 + rank))
    post = sp.sample(world * 256, n_final_samples=world * 256 - 5, sampler_kwargs=dict(n_steps=2, n_final_steps=1),
                     store_sample_history=False)
    res["beta"], res["logz"], res["logz_err"] = np.array(sp.history.beta), float(post.log_evidence), float(post.log_evidence_error)
    res["n_post"] = len(post.x)
    # lagged adaptation over eight ranks (slot layout: the single-rank particle order)
    spl = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=2.0, engine=eng, seed=3), xp=np,
                 engine=eng, comm=comm, rng=np.random.default_rng(4))
    spl.shard_layout = "slots"
    postl = spl.sample(world * 128, sampler_kwargs=dict(n_steps=5, adapt_lag=4, target_acceptance_rate=0.95), store_sample_history=False)
    res["lag_beta"], res["lag_logz"] = np.array(spl.history.beta), float(postl.log_evidence)
    res["lag_rho"], res["lag_acc"] = np.array(spl.history.mcmc_step_size), np.array(spl.history.mcmc_acceptance)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def eight_rank_results(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("gloo8"))
    spawn_ranks(_worker8, 8, lambda port: (8, port, out))
    return [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(8)]


def test_world8_owner_layout_is_generator_choice(eight_rank_results, oracle):
    """Eight ranks: every rank holds exactly the sub-sequence of numpy's Generator.choice index vector that points into
    its shard - first from equal shards, then from the ragged population to a size that 8 does not divide."""
    from conftest import synth

    world, n = 8, 8 * 4096
    x, ll, lp, lq = synth(n, 4, 13)
    rs = eight_rank_results
    w = oracle.normalized_weights(ll, lp, lq, 0.0, 0.08)
    ref = np.random.default_rng(5).choice(n, size=n, replace=True, p=w)
    counts = [int(((ref >= r * 4096) & (ref < (r + 1) * 4096)).sum()) for r in range(world)]
    for r, res in enumerate(rs):
        assert np.array_equal(res["own_x"], x[ref[(ref >= r * 4096) & (ref < (r + 1) * 4096)]])
        assert res["own_counts"].tolist() == counts
        assert res["own_flags"].tolist() == [n, 1, int(sum(counts[:r]))]
        assert float(res["own_var"]) == pytest.approx(oracle.log_evidence_ratio_variance(ll, lp, lq, 0.0, 0.08), rel=1e-10)
        assert np.array_equal(res["fb"], rs[0]["fb"])
    # second step: Generator.choice on the concatenated (rank-major) ragged population, n - 3 draws
    allx = np.concatenate([res["own_x"] for res in rs])
    ll2 = -0.5 * np.sum(allx**2, axis=1)
    lq2 = -0.5 * np.sum((allx / 1.5) ** 2, axis=1) - 4 * np.log(1.5) - 0.5 * 4 * np.log(2 * np.pi)
    w2 = oracle.normalized_weights(ll2, ll2, lq2, 0.08, 0.2)
    ref2 = np.random.default_rng(6).choice(n, size=n - 3, replace=True, p=w2)
    edges = np.concatenate([[0], np.cumsum(counts)])
    for r, res in enumerate(rs):
        mine = ref2[(ref2 >= edges[r]) & (ref2 < edges[r + 1])]
        assert np.array_equal(res["odd_x"], allx[mine])
    assert int(sum(len(res["odd_x"]) for res in rs)) == n - 3 and rs[0]["odd_counts"].tolist() == [len(res["odd_x"]) for res in rs]


def test_world8_lagged_adaptation_equals_single_rank_with_the_same_lag(eight_rank_results):
    """Eight gloo ranks, adapt_lag = 4 on 5-step mutations (a full block and a one-step tail): the single-rank run's bits."""
    sp, post = _single_rank_lagged(8 * 128, 5, 4)
    for r in eight_rank_results:
        assert np.array_equal(r["lag_beta"], np.array(sp.history.beta))
        assert float(r["lag_logz"]) == pytest.approx(float(post.log_evidence), abs=5e-9)
        assert np.array_equal(r["lag_rho"], np.array(sp.history.mcmc_step_size)) and np.array_equal(r["lag_acc"], np.array(sp.history.mcmc_acceptance))


def test_world8_skew_falls_back_to_slots_and_sampler_runs(eight_rank_results, oracle):
    from conftest import synth

    world, n = 8, 8 * 4096
    x, ll, lp, lq = synth(n, 4, 13)
    rs = eight_rank_results
    ll_skew = ll.copy()
    for r in (2, 5):
        ll_skew[r * 4096:(r + 1) * 4096] += 30.0
    w = oracle.normalized_weights(ll_skew, lp, lq, 0.0, 0.5)
    ref = np.random.default_rng(7).choice(n, size=n, replace=True, p=w)
    assert all(res["skew_flags"].tolist() == [n // world, 0] for res in rs)  # equal shards again: the slot layout ran
    assert np.array_equal(np.concatenate([res["skew_x"] for res in rs]), x[ref])  # ... with the reference's ancestors in draw order
    # sampler: the same schedule and evidence on every rank, the odd final population shared out completely
    assert all(np.array_equal(res["beta"], rs[0]["beta"]) and float(res["logz"]) == float(rs[0]["logz"]) for res in rs)
    assert rs[0]["beta"][-1] == 1.0 and sum(int(res["n_post"]) for res in rs) == world * 256 - 5
    assert abs(float(rs[0]["logz"]) - 2.0 * np.log(np.pi)) < 5 * float(rs[0]["logz_err"]) + 0.1


# ---- signal / await_signal: a producer's failure reaches the waiting ranks ------------------------------------------
def _signal_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aspire_amd.comm import TorchDistComm

    comm = TorchDistComm(torch.device("cpu"))
    res = {}
    # 1: the ordinary hand-off; 2: the producer fails - the waiter raises with its message instead of waiting a day
    if rank == 0:
        comm.signal("job")
        comm.signal("job", error=ValueError("training diverged"))
        res["ok"] = 1
    else:
        comm.await_signal("job", timeout_s=60)
        try:
            comm.await_signal("job", timeout_s=60)
            res["raised"] = 0
        except RuntimeError as exc:
            res["raised"] = 1
            res["msg"] = np.frombuffer(str(exc).encode(), dtype=np.uint8)
    np.savez(os.path.join(out_dir, f"sig{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_signal_carries_a_producer_failure_to_the_waiting_ranks(tmp_path):
    spawn_ranks(_signal_worker, 2, lambda port: (2, port, str(tmp_path)))
    r1 = np.load(os.path.join(str(tmp_path), "sig1.npz"))
    assert int(r1["raised"]) == 1 and b"training diverged" in bytes(r1["msg"])
