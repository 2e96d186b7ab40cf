"""The CPU oracle pinned against the real reference's golden vectors (tests/golden/ref_*.npz, made by
oracle/make_golden.py from /root/reference) and against numpy / Random123 known answers."""
import numpy as np
import pytest

from conftest import synth


def test_weights_golden(oracle, golden):
    g = golden["ref_weights"]
    for n, d, seed, b0, b in g["cases"]:
        n, d, seed = int(n), int(d), int(seed)
        x, ll, lp, lq = synth(n, d, seed)
        key = f"n{n}_b{b0}_t{b}"
        lw = oracle.log_weights(ll, lp, lq, b0, b)
        if n <= 2000:
            assert np.array_equal(lw, g[key + "_lw"])
        else:
            assert np.array_equal(lw[::257], g[key + "_lw_stride"])
        assert oracle.logsumexp(oracle.unnormalized_log_weights(ll, lp, lq, b0, b)) == float(g[key + "_lse_unnorm"])
        assert oracle.effective_sample_size(lw) == pytest.approx(float(g[key + "_ess"]), rel=1e-13)
        assert oracle.log_evidence_ratio(ll, lp, lq, b0, b) == float(g[key + "_ratio"])
        assert oracle.log_evidence_ratio_variance(ll, lp, lq, b0, b) == pytest.approx(float(g[key + "_var"]), rel=1e-12)


def test_beta_golden(oracle, golden):
    g = golden["ref_beta"]
    for n, d, seed, b0, tol, ti, b_ref in g["cases"]:
        n, d, seed, ti = int(n), int(d), int(seed), int(ti)
        x, ll, lp, lq = synth(n, d, 0, 2.0) if seed == 0 else synth(n, d, seed)
        r = oracle.determine_beta(ll, lp, lq, b0, beta_tolerance=tol, target_efficiency=[0.5, (0.3, 0.9)][ti])
        assert r.beta == b_ref and not r.stalled
    x, ll, lp, lq = synth(2000, 4, 0, 2.0)
    r = oracle.determine_beta(ll, lp, lq, 0.0, min_beta_step=0.2, max_beta_step=0.25, beta_tolerance=1e-6,
                              adaptive_min_beta_step=True)
    assert [r.beta, r.min_beta_step] == list(g["n2000_minstep"])


def test_survey_anchor_values(oracle):
    # SURVEY.md §8c anchors observed on the reference
    x, ll, lp, lq = synth(2000, 4, 0, 2.0)
    assert oracle.determine_beta(ll, lp, lq, 0.0, beta_tolerance=1e-8).beta == 0.16844338178634644
    assert oracle.determine_beta(ll, lp, lq, 0.0, beta_tolerance=1e-6).beta == 0.1684427261352539


def test_resample_golden(oracle, golden):
    g = golden["ref_resample"]
    for n, d, seed, b0, b, n_out in g["cases"]:
        n, d, seed, n_out = int(n), int(d), int(seed), int(n_out)
        x, ll, lp, lq = synth(n, d, seed)
        st = oracle.pcg64_state_from_numpy(np.random.default_rng(1000 + seed))
        u = oracle.pcg64_random(st, n_out)
        key = f"n{n}_b{b0}_t{b}_o{n_out}"
        assert np.array_equal(oracle.resample_indices(ll, lp, lq, b0, b, u), g[key + "_idx"])
        assert np.array_equal(oracle.pcg64_random(st, 3), g[key + "_next_u"])
    x, ll, lp, lq = synth(2000, 4, 32)
    u = np.random.default_rng(77).random(50)
    assert np.array_equal(oracle.resample_indices(ll, lp, lq, 0.4, 0.4, u, uniform_weights=True), g["samebeta_idx"])


def test_choice_equivalence_live(oracle):
    """numpy Generator.choice(p=w) == cumsum/searchsorted restatement, live."""
    x, ll, lp, lq = synth(5000, 3, 8)
    w = oracle.normalized_weights(ll, lp, lq, 0.0, 0.3)
    for n_out in (5000, 37):
        ref = np.random.default_rng(3).choice(5000, size=n_out, replace=True, p=w)
        u = np.random.default_rng(3).random(n_out)
        assert np.array_equal(oracle.searchsorted_right(oracle.cdf_from_weights(w), u), ref)


def test_pcg64_vs_numpy(oracle):
    for seed in (0, 1, 2**40 + 7):
        rng = np.random.default_rng(seed)
        st = oracle.pcg64_state_from_numpy(rng)
        assert np.array_equal(oracle.pcg64_random(st, 1000), rng.random(1000))
        oracle.pcg64_advance(st, 123456789)
        rng.bit_generator.advance(123456789)
        assert np.array_equal(st, oracle.pcg64_state_from_numpy(rng))


def test_philox_known_answers(oracle):
    """Random123 kat_vectors, philox4x32 10 rounds."""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, out in kat:
        assert list(oracle.philox4x32_10(ctr, key)) == out


def test_nan_guard_and_compaction(oracle):
    with pytest.raises(ValueError, match="Log weights contain NaN"):
        oracle.log_weights([0.0, np.nan], [0.0, 0.0], [0.0, 0.0], 0.0, 0.5)
    x = np.arange(12.0).reshape(6, 2)
    ll = np.array([0, -np.inf, 1, np.nan, 2, 3.0])
    lp = np.array([0, 0, np.inf, 0, 0, 0.0])
    xo, llo, lpo, lqo = oracle.compact_valid(x, ll, lp, np.arange(6.0))
    assert np.array_equal(lqo, [0.0, 4.0, 5.0]) and np.array_equal(xo, x[[0, 4, 5]])


def test_pcn_oracle_is_a_valid_kernel(oracle):
    """The restated pCN spec leaves its target invariant (statistical pin; parity with minipcn is unpinned)."""
    d, n = 3, 20000
    g = np.random.default_rng(0)
    x = g.normal(size=(n, d)) * np.sqrt(0.5)
    tgt = oracle.Mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    q = oracle.Mixture([0.0], np.zeros((1, d)), np.full((1, d), 0.25))
    ll, lp, lq = tgt.logpdf(x), tgt.logpdf(x), q.logpdf(x)
    L = np.diag([0.8, 1.0, 1.3])
    rho = 0.5
    for t in range(15):
        acc = oracle.pcn_step(x, ll, lp, lq, 1.0, np.full(d, 0.1), L, np.linalg.inv(L), rho, tgt, tgt, q, 5, 0, t)
        rho = oracle.pcn_adapt(rho, acc / n, 0.234, t)
    assert np.all(np.abs(x.mean(0)) < 0.03) and np.all(np.abs(x.var(0) - 0.5) < 0.03)
    np.testing.assert_allclose(ll, -0.5 * (x**2).sum(1), rtol=1e-12)


@pytest.mark.parametrize("d,n_layers,hidden", [(32, 4, 64), (4, 2, 32), (20, 3, 64), (64, 2, 128)])
def test_coupling_flow_oracle_vs_torch(oracle, d, n_layers, hidden):
    """orc_coupling_logprob (fp32, C) against the torch CouplingFlow it restates: fp32 module within fp32
    rounding, fp64 module (same parameters) within the fp32 evaluation error."""
    import torch

    from conftest import random_coupling_flow

    flow = random_coupling_flow(d, n_layers, hidden)
    x = np.random.default_rng(5).normal(size=(257, d)) * 1.3
    ws, bs = flow.export_layers()
    got = oracle.coupling_logprob(x, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    ref32 = flow.log_prob(torch.as_tensor(x)).double().numpy()
    np.testing.assert_allclose(got, ref32, rtol=5e-6, atol=2e-4)
    flow64 = random_coupling_flow(d, n_layers, hidden, dtype=torch.float64)
    ref64 = flow64.log_prob(torch.as_tensor(x)).numpy()
    np.testing.assert_allclose(got, ref64, rtol=2e-5, atol=5e-4)


@pytest.mark.parametrize("d,n_tr,hidden", [(32, 3, 64), (7, 2, 32), (20, 3, 64), (1, 2, 32), (48, 2, 128)])
def test_maf_flow_oracle_vs_torch(oracle, d, n_tr, hidden):
    """orc_maf_logprob (fp32, C) against the torch MAFFlow it restates (the reference's default flow class,
    flows/torch/flows.py:140-168; zuko absent: the architecture is this repository's statement): fp32 module within fp32
    rounding, the same parameters evaluated in fp64 within the fp32 evaluation error."""
    import torch

    from conftest import random_maf_flow

    flow = random_maf_flow(d, n_tr, hidden)
    x = flow.loc.numpy() + 1.3 * flow.scale.numpy() * np.random.default_rng(5).normal(size=(257, d))
    ws, bs = flow.export_layers()
    got = oracle.maf_logprob(x, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    ref32 = flow.log_prob(torch.as_tensor(x)).double().numpy()
    np.testing.assert_allclose(got, ref32, rtol=5e-6, atol=2e-4)
    ref64 = flow.log_prob_f64(x).numpy()
    np.testing.assert_allclose(got, ref64, rtol=2e-5, atol=5e-4)
    # the masks are autoregressive: output i of a transform does not move when x_j, j >= i (in the transform's order), does
    z0, _ = flow.forward(torch.as_tensor(x[:4], dtype=torch.float32))
    assert torch.isfinite(z0).all()


@pytest.mark.parametrize("kind", ["coupling", "maf"])
def test_tpcn_flow_step_is_the_composition_of_its_pieces(oracle, kind):
    """orc_tpcn_flow_step_kind - the reference's default pairing, step_fn="tpcn" (smc/minipcn.py:46-49) with a neural proposal
    density inside log p_t (smc/base.py:507-519) - spelled out with the oracle's own pieces: at beta = 1 log q drops out of the
    accept rule, so positions and accept decisions are orc_tpcn_step's bit for bit; at beta < 1 the recorded margins are
    [log p_t(x') + c(q')] - [log p_t(x) + c(q)] - log u with the flow's log q, c = orc_tpcn_corr; accepted rows carry the flow's
    density at the new position; any thread count gives the same bits."""
    from conftest import random_coupling_flow, random_maf_flow

    n, d, nu, rho = 600, 8, 4.5, 0.3
    g = np.random.default_rng(3)
    flow = random_coupling_flow(d, 2, 32) if kind == "coupling" else random_maf_flow(d, 2, 32)
    flp = oracle.coupling_logprob if kind == "coupling" else oracle.maf_logprob
    ws, bs = flow.export_layers()
    loc, scale = flow.loc.numpy(), flow.scale.numpy()
    x0 = 1.2 * g.normal(size=(n, d))
    tgt = oracle.Mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    flat = oracle.Mixture([0.0], np.zeros((1, d)), np.zeros((1, d)))
    ll0, lq0 = tgt.logpdf(x0), flp(x0, ws, bs, loc, scale)
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.tril(np.linalg.cholesky(np.eye(d) + 0.2 * a @ a.T))
    Linv, mu = np.tril(np.linalg.inv(L)), 0.1 * g.normal(size=d)
    # beta = 1: the same decisions and positions as the built-in-density tpCN step
    xa, lla, lpa, lqa = x0.copy(), ll0.copy(), ll0.copy(), lq0.copy()
    acc_a = oracle.tpcn_flow_step(xa, lla, lpa, lqa, 1.0, mu, L, Linv, rho, nu, tgt, tgt, ws, bs, loc, scale, 77, 5, 2, flow_kind=kind)
    xb, llb, lpb, lqb = x0.copy(), ll0.copy(), ll0.copy(), np.zeros(n)
    acc_b = oracle.tpcn_step(xb, llb, lpb, lqb, 1.0, mu, L, Linv, rho, nu, tgt, tgt, flat, 77, 5, 2)
    assert acc_a == acc_b and 0 < acc_a < n and np.array_equal(xa, xb) and np.array_equal(lla, llb)
    moved = np.any(xa != x0, axis=1)
    np.testing.assert_array_equal(lqa[moved], flp(xa[moved], ws, bs, loc, scale))
    assert np.array_equal(lqa[~moved], lq0[~moved])
    # beta < 1: the margin of every particle, recomputed from the pieces (the proposal is the same as at beta = 1)
    beta = 0.35
    x, ll, lp, lq = x0.copy(), ll0.copy(), ll0.copy(), lq0.copy()
    with oracle.accept_margins(n) as m:
        acc = oracle.tpcn_flow_step(x, ll, lp, lq, beta, mu, L, Linv, rho, nu, tgt, tgt, ws, bs, loc, scale, 77, 5, 2, flow_kind=kind)
        m = m.copy()
    # every proposal, from a step that accepts everything finite: flat targets at beta = 1 still use the t correction, so take
    # the proposals from the accepted rows of the two runs and check the others through the margin identity on accepted rows
    acc_rows = np.any(x != x0, axis=1)
    assert acc_rows.sum() == acc and np.array_equal(acc_rows, m > 0)
    y0 = (x0 - mu) @ Linv.T
    y1 = (x - mu) @ Linv.T
    q0, q1 = (y0**2).sum(1), (y1**2).sum(1)
    u = np.array([oracle.pcn_noise(77, 5 + i, 2, d)[1] for i in range(n)])
    lpt_new = np.array([oracle.log_p_t(a_, b_, c_, beta) for a_, b_, c_ in zip(ll, lp, lq)])
    lpt_old = np.array([oracle.log_p_t(a_, b_, c_, beta) for a_, b_, c_ in zip(ll0, ll0, lq0)])
    want = (lpt_new + oracle.tpcn_corr(q1, d, nu)) - (lpt_old + oracle.tpcn_corr(q0, d, nu)) - np.log(u)
    np.testing.assert_allclose(m[acc_rows], want[acc_rows], rtol=1e-9, atol=1e-9)
    # threads
    for nt in (3, 0):
        x2, l2, p2, q2 = x0.copy(), ll0.copy(), ll0.copy(), lq0.copy()
        assert oracle.tpcn_flow_step(x2, l2, p2, q2, beta, mu, L, Linv, rho, nu, tgt, tgt, ws, bs, loc, scale, 77, 5, 2, n_threads=nt,
                                     flow_kind=kind) == acc
        assert np.array_equal(x2, x) and np.array_equal(q2, lq)
    with pytest.raises(ValueError):
        oracle.tpcn_flow_step(x2, l2, p2, q2, beta, mu, L, Linv, rho, 0.0, tgt, tgt, ws, bs, loc, scale, 77, 5, 2, flow_kind=kind)


def test_transforms_golden(oracle, golden):
    """orc_transform against the real reference's CompositeTransform (forward after fit, inverse, log|det J|):
    elementary functions come from libm here and from numpy/scipy there, hence 1e-13 relative."""
    g = golden["ref_transforms"]
    for name in g["names"]:
        kind, per, lo, up = g[f"{name}_kind"], g[f"{name}_periodic"], g[f"{name}_lower"], g[f"{name}_upper"]
        mean = g[f"{name}_mean"] if int(g[f"{name}_affine"]) else None
        std = g[f"{name}_std"] if int(g[f"{name}_affine"]) else None
        z, lj = oracle.transform(g[f"{name}_x"], kind, per, lo, up, mean, std, 1e-6)
        np.testing.assert_allclose(z, g[f"{name}_z"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(z, g[f"{name}_z_fit"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(lj, g[f"{name}_lj"], rtol=1e-13, atol=1e-12)
        x2, lj2 = oracle.transform(g[f"{name}_z2"], kind, per, lo, up, mean, std, 1e-6, inverse=True)
        np.testing.assert_allclose(x2, g[f"{name}_x2"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(lj2, g[f"{name}_lj2"], rtol=1e-13, atol=1e-12)
        # round trip through the oracle itself (periodic dims come back wrapped)
        xb, ljb = oracle.transform(z, kind, per, lo, up, mean, std, 1e-6, inverse=True)
        inside = ~per.astype(bool)
        clamp = (kind != 0)
        ok = inside & ~clamp
        np.testing.assert_allclose(xb[:, ok], g[f"{name}_x"][:, ok], rtol=1e-10, atol=1e-10)


def test_flow_pcn_step_threads_equal_sequential_composition(oracle):
    """orc_pcn_flow_step (bench.py's CPU baseline): any thread count gives the single-thread result, and that result is the
    step spelled out with the oracle's own pieces (proposal of orc_pcn_step, orc_coupling_logprob, accept rule)."""
    from conftest import random_coupling_flow

    n, d = 700, 8
    g = np.random.default_rng(3)
    flow = random_coupling_flow(d, 2, 32)
    ws, bs = flow.export_layers()
    loc, scale = flow.loc.numpy(), flow.scale.numpy()
    x0 = 1.2 * g.normal(size=(n, d))
    tgt = oracle.Mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    ll0 = tgt.logpdf(x0)
    lq0 = oracle.coupling_logprob(x0, ws, bs, loc, scale)
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.linalg.cholesky(np.eye(d) + 0.2 * a @ a.T)
    Linv, mu = np.linalg.inv(L), 0.1 * g.normal(size=d)
    outs = []
    for nt in (1, 3, 0):
        x, ll, lp, lq = x0.copy(), ll0.copy(), ll0.copy(), lq0.copy()
        acc = oracle.pcn_flow_step(x, ll, lp, lq, 0.4, mu, np.tril(L), np.tril(Linv), 0.3, tgt, tgt, ws, bs, loc, scale, 77, 5, 2,
                                   n_threads=nt)
        outs.append((acc, x, ll, lq))
    for acc, x, ll, lq in outs[1:]:
        assert acc == outs[0][0] and np.array_equal(x, outs[0][1]) and np.array_equal(lq, outs[0][3])
    acc, x, ll, lq = outs[0]
    assert 0 < acc < n
    # accepted rows carry the flow's log-density at the new position, rejected rows are untouched
    moved = np.any(x != x0, axis=1)
    assert moved.sum() == acc
    np.testing.assert_array_equal(lq[moved], oracle.coupling_logprob(x[moved], ws, bs, loc, scale))
    assert np.array_equal(lq[~moved], lq0[~moved]) and np.array_equal(ll[moved], tgt.logpdf(x[moved]))
    # the proposals are orc_pcn_step's: a step with a flat log q accepts a superset-independent check of positions
    x2, l2 = x0.copy(), ll0.copy()
    flat = oracle.Mixture([0.0], np.zeros((1, d)), np.zeros((1, d)))
    oracle.pcn_step(x2, l2, ll0.copy(), np.zeros(n), 0.4, mu, np.tril(L), np.tril(Linv), 0.3, tgt, tgt, flat, 77, 5, 2)
    both = moved & np.any(x2 != x0, axis=1)
    assert both.sum() > 10 and np.array_equal(x[both], x2[both])
    assert oracle.max_threads() >= 1
