"""Masked autoregressive flow on the HIP kernels (include/asmc.h ASMC_FLOW_MAF).

The reference's default flow class is zuko's MAF (`/root/reference/src/aspire/flows/torch/flows.py:140-168`), evaluated by
`log_prob` (:368-387) in every MCMC step and by `sample_and_log_prob` (:327-346) in the proposal draw.  zuko is absent from the
image: the arithmetic is this repository's statement of the architecture (`aspire_amd/flows.py` MAFFlow; parity unpinned), so the
kernels are checked against (i) the SAME parameters evaluated in fp64 by the torch modules - the north star's bar, 1e-6 relative
on log-weights - and (ii) the C oracle's fp32 restatement (`orc_maf_logprob`, `orc_pcn_flow_step_kind`).
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def random_maf(d, n_transforms=3, hidden=64, seed=3):
    from conftest import random_maf_flow

    return random_maf_flow(d, n_transforms, hidden, seed)


@pytest.fixture(scope="module")
def eng():
    from aspire_amd.engine import HipEngine

    return HipEngine(0, n_max=1 << 20, d_max=32)


@pytest.fixture(scope="module")
def oracle():
    import oracle as O

    return O


@pytest.mark.parametrize("d,n_tr,hidden,n,dtype", [(32, 3, 64, 70001, torch.float64), (32, 3, 64, 4097, torch.float32),
                                                   (7, 2, 32, 3000, torch.float64), (16, 3, 64, 5000, torch.float64),
                                                   (20, 2, 64, 2500, torch.float64), (32, 1, 128, 1500, torch.float64),
                                                   (32, 3, 128, 3000, torch.float64), (9, 3, 128, 1000, torch.float32),  # (round 5: "do not fit in LDS together")
                                                   (1, 2, 32, 100, torch.float64), (32, 3, 64, 1, torch.float64)])
def test_maf_logprob_vs_fp64_and_oracle(eng, oracle, d, n_tr, hidden, n, dtype):
    """asmc_coupling_logprob with kind = ASMC_FLOW_MAF: the same flow in fp64 to 1e-6 relative (the north star's bar), the C
    oracle's fp32 restatement to fp32 rounding; odd and padded dims, a ragged last tile, float32 rows."""
    flow = random_maf(d, n_tr, hidden)
    dev = flow.device_coupling(eng)
    assert dev.kind == 1
    g = np.random.default_rng(5)
    x = (flow.loc.numpy() + 1.2 * flow.scale.numpy() * g.normal(size=(n, d))).astype(np.float64)
    xd = torch.as_tensor(x, device=eng.device).to(dtype).contiguous()
    got = eng.coupling_logprob(xd, dev).cpu().numpy()
    xr = xd.double().cpu().numpy()
    ws, bs = flow.export_layers()
    want = oracle.maf_logprob(xr, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    assert np.all(np.isfinite(got))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=3e-4)
    ref64 = flow.log_prob_f64(xr).numpy()
    rel = np.abs(got - ref64) / np.maximum(np.abs(ref64), 1.0)
    assert rel.max() <= 1e-6, rel.max()


def test_maf_logprob_fp32_mfma_chain_agrees(eng, monkeypatch):
    """ASMC_FLOW_MATH=f32 runs the same transforms on the fp32-input MFMA: another instruction stream, the same numbers to fp32."""
    flow = random_maf(32, 3, 64, seed=9)
    dev = flow.device_coupling(eng)
    x = eng.asarray(np.random.default_rng(1).normal(size=(20000, 32)))
    a = eng.coupling_logprob(x, dev).cpu().numpy()
    monkeypatch.setenv("ASMC_FLOW_MATH", "f32")
    b = eng.coupling_logprob(x, dev).cpu().numpy()
    assert not np.array_equal(a, b)
    ref = flow.log_prob_f64(x).cpu().numpy()
    for v in (a, b):
        assert np.max(np.abs(v - ref) / np.maximum(np.abs(ref), 1.0)) <= 1e-6


def test_maf_out_of_range_operand_is_nan_not_a_wrong_number(eng):
    """An activation beyond the fp16 operand range of the split products must surface as NaN (rejected and counted by the
    mutation), never as a finite wrong density."""
    flow = random_maf(32, 3, 64, seed=2)
    dev = flow.device_coupling(eng)
    x = np.random.default_rng(0).normal(size=(256, 32))
    x[7] *= 1e6
    got = eng.coupling_logprob(eng.asarray(x), dev).cpu().numpy()
    assert np.isnan(got[7]) and np.all(np.isfinite(np.delete(got, 7)))


@pytest.mark.parametrize("d,n_tr,hidden,n,dtype", [(32, 3, 64, 50000, torch.float64), (7, 2, 32, 4000, torch.float32),
                                                   (16, 3, 64, 4097, torch.float64),
                                                   (32, 3, 128, 20000, torch.float64), (12, 2, 128, 3000, torch.float64)])  # (round 6: the streamed layout at d <= 32)
def test_maf_sample_inverts_the_density_pass(eng, d, n_tr, hidden, n, dtype):
    """asmc_coupling_sample with kind = ASMC_FLOW_MAF (d passes per transform, whatever the variable order): the returned log q
    is the density of the returned x (fp64 evaluation of the same flow), pushing x back through the flow recovers a standard
    normal latent, and a draw does not depend on how the population is sharded (global particle index)."""
    flow = random_maf(d, n_tr, hidden, seed=4)
    dev = flow.device_coupling(eng)
    x, lq = eng.coupling_sample(n, dtype, dev, 1234, 0, 1)
    assert torch.isfinite(x).all() and torch.isfinite(lq).all()
    ref = flow.log_prob_f64(x.double()).to(lq.device)
    tol = 2e-6 if dtype == torch.float64 else 2e-4  # (float32 rows: x itself is rounded after the density was formed)
    rel = ((lq - ref).abs() / ref.abs().clamp_min(1.0)).max()
    assert float(rel) <= tol, float(rel)
    z, _ = flow.forward(x.float().cpu())
    z = z.double().numpy()
    assert abs(z.mean()) < 5.0 / math.sqrt(n * d) and abs(z.var() - 1.0) < 0.05
    # the second shard of a two-rank run draws the rows the one-rank run has there
    h = n // 2
    x2, lq2 = eng.coupling_sample(n - h, dtype, dev, 1234, h, 1)
    assert torch.equal(x2, x[h:]) and torch.equal(lq2, lq[h:])
    x3, _ = eng.coupling_sample(n, dtype, dev, 1234, 0, 2)  # another draw id: another stream
    assert not torch.equal(x3, x)


@pytest.mark.parametrize("d,n_tr,hidden,n,weak", [(32, 3, 64, 30000, False), (32, 3, 64, 30000, True), (7, 2, 32, 3000, True)])
def test_maf_sample_fixed_point_exit_returns_the_d_pass_bits(eng, monkeypatch, d, n_tr, hidden, n, weak):
    """k_maf_sample stops inverting a transform once a pass returns its input bit for bit (round 5): the rows and log q are the
    bits of the full d-pass loop (ASMC_MAF_SAMPLE_ALL_PASSES=1), for a flow with strong couplings (where a transform needs its d
    passes anyway) and for a weakly coupled one (the shape of a trained, near-identity flow), which finishes in a few passes."""
    flow = random_maf(d, n_tr, hidden, seed=14)
    if weak:
        from aspire_amd.flows import _MaskedLinear

        with torch.no_grad():
            for layer in flow.layers:
                last = [m for m in layer.net if isinstance(m, _MaskedLinear)][-1]
                last.weight.mul_(0.02)
        flow._version += 1
    dev = flow.device_coupling(eng)
    x, lq = eng.coupling_sample(n, torch.float64, dev, 77, 0, 3)
    monkeypatch.setenv("ASMC_MAF_SAMPLE_ALL_PASSES", "1")
    x2, lq2 = eng.coupling_sample(n, torch.float64, dev, 77, 0, 3)
    assert torch.equal(x, x2) and torch.equal(lq, lq2)
    assert torch.isfinite(lq).all()


def test_maf_sample_flags_a_transient_overflow(eng):
    """an intermediate pass whose hidden activations leave the fp16 operand range (a not-yet-final coordinate times a
    large first-layer weight) poisons coordinates that were already final through 0 x inf; the range check now covers EVERY pass,
    so such a row comes back with log q = NaN instead of a silently clamped position."""
    from aspire_amd.flows import _MaskedLinear

    d = 8
    flow = random_maf(d, 1, 32, seed=5)
    with torch.no_grad():
        first = [m for m in flow.layers[0].net if isinstance(m, _MaskedLinear)][0]
        last = [m for m in flow.layers[0].net if isinstance(m, _MaskedLinear)][-1]
        first.weight.mul_(400.0)   # hidden pre-activations of garbage inputs leave the fp16 range ...
        last.weight.mul_(30.0)     # ... and the shifts t are large enough to produce such inputs
    flow._version += 1
    dev = flow.device_coupling(eng)
    x, lq = eng.coupling_sample(4096, torch.float64, dev, 5, 0, 1)
    ok = torch.isfinite(lq)
    # rows that came back finite are exact: density of the returned rows in fp64
    if ok.any():
        ref = flow.log_prob_f64(x[ok]).to(lq.device)
        good = torch.isfinite(ref)
        rel = ((lq[ok][good] - ref[good]).abs() / ref[good].abs().clamp_min(1.0))
        assert float(rel.max()) <= 1e-4, float(rel.max())
    assert torch.isfinite(x).all()


def _mutation_setup(eng, n, d, seed):
    g = np.random.default_rng(seed)
    x0 = 0.9 * g.normal(size=(n, d))
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.linalg.cholesky(0.8 * (np.eye(d) + 0.2 * a @ a.T))
    return x0, 0.05 * g.normal(size=d), np.tril(L), np.tril(np.linalg.inv(L))


@pytest.mark.parametrize("d,hidden,n,fused,nu", [(32, 64, 6000, True, 0.0), (32, 32, 2500, True, 0.0), (16, 64, 3000, True, 0.0),
                                                 (7, 32, 2000, True, 0.0), (20, 64, 2500, True, 0.0), (31, 32, 2000, True, 0.0),
                                                 (3, 64, 1500, True, 0.0), (16, 128, 1500, False, 0.0), (32, 128, 2000, False, 0.0),
                                                 (32, 64, 6000, True, 5.0), (20, 64, 2500, True, 3.0), (7, 32, 2000, True, 8.0),
                                                 (16, 128, 1500, False, 5.0)])
def test_maf_mutation_vs_oracle(eng, oracle, d, hidden, n, fused, nu):
    """asmc_pcn_mutate_flow with an autoregressive proposal density against the oracle's restatement of the whole step
    (orc_pcn_flow_step_kind, flow_kind = maf): every d <= 32 (odd ones too) takes the ONE-kernel step (k_pcn_flow_fused, the MAF
    instantiation; d < 32 zero-padded inside the library); hidden width 128 (round 5: one transform at most, on a three-kernel
    loop - more did not fit the LDS) lives in the streamed-weight layout since round 6 and takes the one-kernel step of the
    16-particle-group kernels, zero-padded to D = 64 - no torch op, no host round trip inside the loop either way.
    nu > 0: the reference's DEFAULT pairing - step_fn="tpcn" (smc/minipcn.py:46-49) with flow_class="MAF"
    (flows/torch/flows.py:140) - against orc_tpcn_flow_step_kind."""
    n_steps, beta, rho = 3, 0.4, 0.35
    flow = random_maf(d, 3, hidden, seed=6)
    dev = flow.device_coupling(eng)
    assert eng.lib.asmc_flow_layout(dev.kind, d, hidden) == (1 if hidden == 128 else 0)
    ws, bs = flow.export_layers()
    x0, mu, L, Linv = _mutation_setup(eng, n, d, 8)
    tgt_o = oracle.Mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    t_ll = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    xr, llr = x0.copy(), tgt_o.logpdf(x0)
    lpr, lqr = llr.copy(), oracle.maf_logprob(x0, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    xd = eng.asarray(x0)
    lld, lpd, lqd = eng.asarray(llr), eng.asarray(lpr), eng.coupling_logprob(xd, dev)
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate_flow(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), t_ll, t_ll, dev,
                                      4242, 17, rho, n_steps, 9, 0.234, False, "f64", nu)
    rep = eng.profile_report()
    eng.profile(False)
    if fused:
        assert rep["k_pcn_flow_fused"][0] == n_steps and "k_maf_logprob" not in rep
    else:
        assert rep["k_tpcn_flow16" if nu > 0 else "k_pcn_flow16"][0] == n_steps and "k_pcn_flow_fused" not in rep and "k_maf_logprob" not in rep, sorted(rep)
    acc_ref, margins = [], []
    for t in range(n_steps):
        with oracle.accept_margins(n) as m:
            if nu > 0:
                acc_ref.append(oracle.tpcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, tgt_o, tgt_o, ws, bs,
                                                     flow.loc.numpy(), flow.scale.numpy(), 4242, 17, 9 + t, "f64", 0, flow_kind="maf"))
            else:
                acc_ref.append(oracle.pcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, tgt_o, tgt_o, ws, bs, flow.loc.numpy(),
                                                    flow.scale.numpy(), 4242, 17, 9 + t, "f64", 0, flow_kind="maf"))
        margins.append(m.copy())
    got = xd.cpu().numpy()
    close = np.all(np.abs(got - xr) <= 1e-9 * (1 + np.abs(xr)), axis=1)
    assert (~close).sum() <= 12, (~close).sum()
    razor = np.min(np.abs(np.array(margins)), axis=0)  # rows that ended elsewhere took their other decision at a razor's edge
    assert np.all(razor[~close] <= 1e-4), razor[~close]
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= 12) and 0.05 < np.mean(n_acc) / n < 0.95
    np.testing.assert_allclose(lld.cpu().numpy()[close], llr[close], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lqd.cpu().numpy()[close], lqr[close], rtol=1e-5, atol=3e-4)
    # carried log q = the flow's density at the returned positions
    torch.testing.assert_close(lqd, eng.coupling_logprob(xd, dev), rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("noise,nu", [("f64", 0.0), ("f32", 0.0), ("f64", 5.0)])
def test_maf_fused_step_vs_split_kernels(eng, noise, nu):
    """The fused MAF step against the three-kernel device loop (ASMC_FLOW_SPLIT=1: propose / k_maf_logprob / accept) on the same
    counters: same proposals, the same tile code for the flow - positions agree to the rounding of the whitened state, accept
    decisions differ only at razor edges; the fp32 noise generator and the Student-t reference included; ragged last tile."""
    n, d, n_steps, beta, rho = 70001, 32, 4, 0.35, 0.4
    flow = random_maf(d, 3, 64, seed=12)
    dev = flow.device_coupling(eng)
    x0, mu, L, Linv = _mutation_setup(eng, n, d, 3)
    t_ll = eng.make_mixture([0.3], np.full((1, d), 0.25), np.ones((1, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu_d, L_d, Li_d = eng.asarray(mu), eng.asarray(L), eng.asarray(Linv)

    def run(split):
        x = eng.asarray(x0)
        ll, lp, lq = eng.mixture_logpdf(x, t_ll), eng.mixture_logpdf(x, t_lp), eng.coupling_logprob(x, dev)
        if split:
            os.environ["ASMC_FLOW_SPLIT"] = "1"
        try:
            eng.profile(True)
            acc, _, _ = eng.pcn_mutate_flow(x, ll, lp, lq, beta, mu_d, L_d, Li_d, t_ll, t_lp, dev, 77, 1000, rho, n_steps, 5, 0.234,
                                            False, noise, nu)
            rep = eng.profile_report()
            eng.profile(False)
        finally:
            os.environ.pop("ASMC_FLOW_SPLIT", None)
        return x, ll, lp, lq, np.asarray(acc), rep

    xa, lla, lpa, lqa, acc_a, rep_a = run(False)
    xb, llb, lpb, lqb, acc_b, rep_b = run(True)
    assert rep_a["k_pcn_flow_fused"][0] == n_steps and "k_maf_logprob" not in rep_a
    assert rep_b["k_maf_logprob"][0] == n_steps and "k_pcn_flow_fused" not in rep_b
    close = ((xa - xb).abs() <= 1e-9 * (1 + xb.abs())).all(dim=1)
    assert int((~close).sum()) <= 8, int((~close).sum())
    assert np.all(np.abs(acc_a - acc_b) <= 8) and 0 < acc_a.sum() < n * n_steps
    torch.testing.assert_close(lqa, eng.coupling_logprob(xa, dev), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(lla, eng.mixture_logpdf(xa, t_ll), rtol=1e-9, atol=1e-9)


def test_maf_fused_step_is_repeatable(eng):
    """Same inputs, same bits, call after call (register spills around the hand-placed MFMA sequences once made the W = 128
    coupling instantiation run-to-run different: tools/stress_fused.py)."""
    n, d = 40000, 32
    flow = random_maf(d, 3, 64, seed=21)
    dev = flow.device_coupling(eng)
    x0, mu, L, Linv = _mutation_setup(eng, n, d, 5)
    t = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    outs = []
    for _ in range(6):
        x = eng.asarray(x0)
        ll, lp, lq = eng.mixture_logpdf(x, t), eng.mixture_logpdf(x, t), eng.coupling_logprob(x, dev)
        eng.pcn_mutate_flow(x, ll, lp, lq, 0.5, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), t, t, dev, 5, 0, 0.3, 6, 0, 0.234,
                            True, "f64", 0.0)
        outs.append((x.clone(), lq.clone()))
        eng.coupling_logprob(eng.asarray(np.random.default_rng(1).normal(size=(3000, d))), dev)  # other kernels in between
    for xo, lqo in outs[1:]:
        assert torch.equal(xo, outs[0][0]) and torch.equal(lqo, outs[0][1])


def test_smc_run_with_the_default_flow_class_stays_on_the_device(eng):
    """`Aspire(flow_backend="maf")`-style run (the reference's DEFAULT flow class) at d = 32: trained MAFFlow as the proposal,
    the initial draw in k_maf_sample, every mutation step in k_pcn_flow_fused - no k_pcn_propose / k_pcn_accept halves with a
    torch log_prob between them - and log Z within 3 sigma of the closed form."""
    from aspire_amd.flows import MAFFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 32, 200_000
    flow = MAFFlow(d, n_transforms=3, hidden_features=(64, 64), seed=7, device=eng.device, dtype=torch.float32)
    flow.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=6)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(11),
                dtype="float64")
    eng.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=16, step_fn="pcn"), store_sample_history=False)
    rep = eng.profile_report()
    eng.profile(False)
    temps = len(sp.history.beta)
    assert "flow: device-side step loop" in sp.last_mutation_path
    assert rep["k_pcn_flow_fused"][0] == 16 * temps and rep["k_maf_sample"][0] >= 1
    for name in rep:
        assert not name.startswith(("k_pcn_propose", "k_pcn_accept", "k_pcn_flow_propose", "k_pcn_flow_accept", "k_maf_logprob",
                                    "k_pad_rows", "k_copy_flagged_rows")), name
    z = (float(out.log_evidence) - 0.5 * d * math.log(math.pi)) / float(out.log_evidence_error)
    assert abs(z) < 3.0, z


def test_smc_run_with_an_autoregressive_flow_of_hidden_width_128_at_d_below_32(eng):
    """`Aspire(flow_backend="zuko", hidden_features=(128, 128))`-style run at d = 16 (the reference forwards any width,
    flows/torch/flows.py:164): round 5 refused more than one such transform on the device; since round 6 the flow lives in the
    streamed-weight layout (asmc_flow_layout = 1) - the draw in k_flow16_sample, every mutation step ONE kernel (k_pcn_flow16 on the
    problem zero-padded to D = 64), on a context created for d_max = 32 - and log Z comes out within 3 sigma of the closed form."""
    from aspire_amd.engine import HipEngine
    from aspire_amd.flows import MAFFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 16, 100_000
    small = HipEngine(0, n_max=n, d_max=32)
    flow = MAFFlow(d, n_transforms=3, hidden_features=(128, 128), seed=7, device=small.device, dtype=torch.float32)
    flow.fit(1.5 * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=6)
    assert small.lib.asmc_flow_layout(1, d, 128) == 1
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=small, rng=np.random.default_rng(11), dtype="float64")
    small.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=8, step_fn="pcn"), store_sample_history=False)
    rep = small.profile_report()
    small.profile(False)
    temps = len(sp.history.beta)
    assert "flow: device-side step loop" in sp.last_mutation_path, sp.last_mutation_path
    assert rep["k_pcn_flow16"][0] == 8 * temps and rep["k_flow16_sample"][0] >= 1, sorted(rep)
    for name in rep:
        assert not name.startswith(("k_pcn_propose", "k_pcn_accept", "k_pcn_flow_propose", "k_pcn_flow_accept", "k_maf_logprob", "k_copy_flagged_rows")), name
    z = (float(out.log_evidence) - 0.5 * d * math.log(math.pi)) / float(out.log_evidence_error)
    assert abs(z) < 3.0, z
    small.close()
