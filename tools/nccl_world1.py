#!/usr/bin/env python
"""The collectives of the sharded path over the REAL RCCL backend with a one-rank world (a single GPU cannot host two
RCCL ranks): device-tensor all-gathers inside the beta search, the rank-total exchange of owner-layout resampling and
the stream-ordered accept-count all-reduce hook, each checked against the collective-free single-rank path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from conftest import synth

    from aspire_amd import smc_math
    from aspire_amd.comm import Comm, TorchDistComm
    from aspire_amd.engine import HipEngine

    n, d = 1_000_000, 32
    eng = HipEngine(0, n_max=n, d_max=d)
    comm = TorchDistComm(eng.device)
    assert not comm._stage and comm.world == 1
    x, ll, lp, lq = synth(n, d, 3)
    xd, lld, lpd, lqd = (eng.asarray(a) for a in (x, ll, lp, lq))
    # 1. beta search: reduce -> all_gather_into_tensor (RCCL) -> decide, all on the stream
    one = eng.find_beta(lld, lpd, lqd, 0.0, 0.5, 1e-6)
    shd = smc_math.find_beta_sharded(eng, comm, lld, lpd, lqd, 0.0, 0.5, 1e-6, n)
    assert shd[0] == one[0] and shd[2] and shd[3] == one[3], (one, shd)
    np.testing.assert_allclose(shd[5], one[5], rtol=1e-11)
    # 2. owner-layout resampling over RCCL all-gathers (partials, tile records, chain states, counts): the slot path's ancestors
    beta = one[0]
    st = smc_math.Stats(*one[5], n)
    idx_o, var_o, _, cnt = smc_math.resample_owner(eng, comm, lld, lpd, lqd, 0.0, beta, n, np.random.default_rng(5), st=st)
    assert cnt == [n]
    var_s, s1p = smc_math.evidence_variance_and_lse(eng, Comm(), lld, lpd, lqd, 0.0, beta, st)
    idx_s, _ = smc_math.resample_indices(eng, Comm(), lld, lpd, lqd, 0.0, beta, n, np.random.default_rng(5), st=st, s1p=s1p)
    # one rank owns everything: the owner layout (tile records -> chain rounds -> ordered selection) must reproduce the
    # single-rank index vector itself, in draw order
    assert np.array_equal(idx_o.cpu().numpy(), idx_s.cpu().numpy())
    assert var_o == var_s
    # 2b. the same step as ONE chain of launches and RCCL collectives (smc_math.shard_step_enqueue: beta*, the weight sums and
    #     the shares stay on the device between the phases; one synchronisation): same beta*, same rows, same variance
    from aspire_amd.samples import SMCSamples

    hc = TorchDistComm(eng.device)
    hc.force_sharded = True
    pop = SMCSamples(x=xd, log_likelihood=lld, log_prior=lpd, log_q=lqd, beta=0.0, xp=torch, engine=eng, comm=hc)
    rng = np.random.default_rng(5)
    eng.profile(True)
    assert pop.speculate_importance_step(0.5, 1e-6, rng)
    rep = eng.profile_report()
    eng.profile(False)
    spec = pop._spec
    assert spec["found"] and spec["beta"] == beta and spec["counts"] == [n], (spec["found"], spec["beta"], beta, spec["counts"])
    assert "k_weights_m2_lse_shard" in rep and "k_weights_map_shard" in rep and "k_bis_decide" not in rep and "k_pack_records" not in rep, sorted(rep)
    rows_s = eng.gather(idx_s, xd, lld, lpd, lqd)
    assert all(torch.equal(a, b) for a, b in zip(spec["rows"], rows_s))
    mean_u = st.S1 / st.n
    assert float(spec["m2"] / st.n / (st.n * mean_u**2)) == var_s
    assert smc_math.pcg64_state(rng).tolist() == smc_math.pcg64_state(np.random.default_rng(5)).tolist()  # untouched until resample()
    # 3. accept-count hook: in-place RCCL all-reduce of the device cell between a step and its adaptation
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))

    def run(hook):
        xx = xd.clone()
        a_, b_, c_ = eng.mixture_logpdf(xx, tgt), eng.mixture_logpdf(xx, tgt), eng.mixture_logpdf(xx, q)
        if hook:
            hc = TorchDistComm(eng.device)
            hc.force_sharded = True  # one real rank: the global sum is the local count
            hc._rccl = None  # this leg: the Python callback form of the hook (torch.distributed's all-reduce)
            eng.set_count_hook(hc, n)
        try:
            return eng.pcn_mutate(xx, a_, b_, c_, 0.5, mu, eye, eye, tgt, tgt, q, 7, 0, 0.3, 12, 0, 0.234, True, "f32"), xx
        finally:
            eng.set_count_hook(None, None)

    (acc0, hist0, rho0), x0 = run(False)
    (acc1, hist1, rho1), x1 = run(True)
    assert np.array_equal(acc0, acc1) and np.array_equal(hist0, hist1) and rho0 == rho1 and torch.equal(x0, x1)
    # 4. fused flow steps (k_pcn_flow_fused): the sharded form leaves the rank's count in the cell, the exchange runs between
    #    the steps and the NEXT step's prologue adapts - with the exchange as the Python callback and as the library's own
    #    ncclAllReduce on a communicator made by comm.rccl_direct(); both must reproduce the single-rank run to the bit
    from conftest import random_coupling_flow

    flow = random_coupling_flow(d, 4, 64)
    dev = flow.device_coupling(eng)
    A = np.eye(d) + 0.05 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    Lf, Lfi = eng.asarray(A), eng.asarray(np.linalg.inv(A))

    def run_flow(kind, adapt=True, steps=9):
        xx = xd.clone()
        a_, b_, c_ = eng.mixture_logpdf(xx, tgt), eng.mixture_logpdf(xx, tgt), eng.coupling_logprob(xx, dev)
        if kind:
            hc = TorchDistComm(eng.device)
            hc.force_sharded = True
            if kind == "python":
                hc._rccl = None
            else:
                assert hc.rccl_direct() is not None
            eng.set_count_hook(hc, n)
        eng.profile(True)
        try:
            out = eng.pcn_mutate_flow(xx, a_, b_, c_, 0.5, mu, Lf, Lfi, tgt, tgt, dev, 11, 0, 0.3, steps, 3, 0.234, adapt, "f64", 0.0)
        finally:
            eng.set_count_hook(None, None)
            rep = eng.profile_report()
            eng.profile(False)
        assert rep["k_pcn_flow_fused"][0] == steps
        assert rep.get("k_pcn_adapt", (0, 0))[0] == (1 if kind else 0), rep.get("k_pcn_adapt")  # only the call's last step
        return out, xx, a_, c_

    for adapt in (True, False):
        (a0, h0, r0), x0, l0, q0 = run_flow(None, adapt)
        assert (len(set(h0.tolist())) > 1) == adapt
        for kind in ("python", "rccl"):
            (a1, h1, r1), x1, l1, q1 = run_flow(kind, adapt)
            assert np.array_equal(a0, a1) and np.array_equal(h0, h1) and r0 == r1, (kind, a0, a1, h0, h1, r0, r1)
            assert torch.equal(x0, x1) and torch.equal(l0, l1) and torch.equal(q0, q1), kind
    (a0, h0, r0), x0, _, _ = run_flow(None, True, steps=1)  # a one-step call: no prologue ever adapts
    (a1, h1, r1), x1, _, _ = run_flow("rccl", True, steps=1)
    assert np.array_equal(a0, a1) and np.array_equal(h0, h1) and r0 == r1 and torch.equal(x0, x1)
    # 5. reference fit: column sums and centred Gram matrix summed over the ranks by the library's own all-reduces
    hc = TorchDistComm(eng.device)
    hc.force_sharded = True
    assert eng.mean_gram_across_ranks_ok(xd, hc)
    s0, g0 = eng.mean_gram(xd, n)
    s1, g1 = eng.mean_gram(xd, n, hc)
    assert np.array_equal(s0, s1) and np.array_equal(g0, g1)
    # 6. the whole sampler through the sharded code path (every `comm.sharded` branch: sharded search, owner-layout
    #    resampling, summed moments, exchanged accept counts with the prologue adaptation) against the single-rank run:
    #    one rank owns everything, so the schedule and the final particles must be identical (log Z to rounding)
    from aspire_amd.flows import CouplingFlow, GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    cflow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=5)
    cflow.fit(1.35 * np.random.default_rng(3).normal(size=(4000, d)), n_epochs=3)

    def run_sampler(sharded, flow, step_fn):
        flow._draws = flow._hip_draws = 0  # the proposal draws are keyed by a per-flow call counter: same draws in both runs
        kw = {}
        if sharded:
            c = TorchDistComm(eng.device)
            c.force_sharded = True
            kw["comm"] = c
        sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(21),
                    dtype="float64", **kw)
        out = sp.sample(200_000, sampler_kwargs=dict(n_steps=4, step_fn=step_fn), store_sample_history=False)
        return sp, out

    for flow, step_fn in ((cflow, "pcn"), (GaussianFlow(d, sigma=1.5, seed=2, engine=eng), "tpcn")):
        sp0, out0 = run_sampler(False, flow, step_fn)
        sp1, out1 = run_sampler(True, flow, step_fn)
        assert sp0.history.beta == sp1.history.beta, (sp0.history.beta, sp1.history.beta)
        assert sp0.history.mcmc_acceptance == sp1.history.mcmc_acceptance
        # (the evidence terms come from different reduction orders - the persistent single-rank kernel against the per-round
        # records of the sharded search - so log Z agrees to rounding, not to the bit; schedule and particles are exact)
        assert abs(float(out0.log_evidence) - float(out1.log_evidence)) <= 1e-13 * abs(float(out0.log_evidence)), (
            float(out0.log_evidence), float(out1.log_evidence))
        x0_, x1_ = (o.x.cpu().numpy() if isinstance(o.x, torch.Tensor) else np.asarray(o.x) for o in (out0, out1))
        assert np.array_equal(x0_, x1_)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("nccl world-1 checks ok: beta*", one[0], "accept", acc1[:3].tolist(), "flow accept", a1.tolist())


if __name__ == "__main__":
    main()
