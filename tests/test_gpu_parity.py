"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference goldens.

Bars: bit-exact for integer/index work (resample indices, PCG64 doubles, gather, compaction, beta*),
relative 1e-12 (far inside the north-star's 1e-6) for log-weights and reductions.
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import synth

from aspire_amd.comm import Comm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(hip_engine):
    return hip_engine


def dev(eng, *arrs):
    return tuple(eng.asarray(a) for a in arrs)


# ---- K1/K2/K4: weights, LSE, ESS, evidence ----------------------------------------------------
@pytest.mark.parametrize("n,d,seed", [(10, 2, 11), (2000, 4, 12), (65536, 8, 13), (1000003, 4, 14)])
def test_weights_stats_vs_oracle(eng, oracle, n, d, seed):
    from aspire_amd import smc_math
    from aspire_amd.comm import Comm

    x, ll, lp, lq = synth(n, d, seed)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    for beta0 in (0.0, 0.3):
        betas = [beta0 + 0.01, 0.5, 1.0]
        stats = smc_math.global_stats(eng, Comm(), lld, lpd, lqd, beta0, betas, n)
        for b, st in zip(betas, stats):
            lw = oracle.unnormalized_log_weights(ll, lp, lq, beta0, b)
            assert st.m == lw.max()  # max is order independent: exact
            assert smc_math.log_evidence_ratio(st) == pytest.approx(oracle.log_evidence_ratio(ll, lp, lq, beta0, b), rel=1e-12, abs=1e-12)
            assert smc_math.ess(st) == pytest.approx(oracle.ess_at_beta(ll, lp, lq, beta0, b), rel=1e-11)
            var = smc_math.evidence_variance(eng, Comm(), lld, lpd, lqd, beta0, b, st)
            assert var == pytest.approx(oracle.log_evidence_ratio_variance(ll, lp, lq, beta0, b), rel=1e-9)


def test_weights_golden_reference(eng, golden):
    from aspire_amd.samples import SMCSamples

    g = golden["ref_weights"]
    for n, d, seed, b0, b in g["cases"]:
        n, d, seed = int(n), int(d), int(seed)
        x, ll, lp, lq = synth(n, d, seed)
        s = SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=b0, engine=eng)
        key = f"n{n}_b{b0}_t{b}"
        lw = s.log_weights(b)
        if n <= 2000:
            np.testing.assert_allclose(lw, g[key + "_lw"], rtol=1e-12, atol=1e-12)
        else:
            np.testing.assert_allclose(lw[::257], g[key + "_lw_stride"], rtol=1e-12, atol=1e-12)
        assert s.effective_sample_size(b) == pytest.approx(float(g[key + "_ess"]), rel=1e-11)
        assert s.log_evidence_ratio(b) == pytest.approx(float(g[key + "_ratio"]), rel=1e-12, abs=1e-12)
        assert s.log_evidence_ratio_variance(b) == pytest.approx(float(g[key + "_var"]), rel=1e-9)


def test_all_beta_buckets_agree(eng, oracle):
    """K = 1..32 candidates in one pass give the same numbers as K separate passes."""
    x, ll, lp, lq = synth(30011, 4, 5)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    betas = np.linspace(0.01, 1.0, 32)
    full = eng.weights_stats(lld, lpd, lqd, 0.0, betas)
    for K in (1, 2, 3, 5, 8, 15, 16, 31):
        part = eng.weights_stats(lld, lpd, lqd, 0.0, betas[:K])
        np.testing.assert_allclose(part[:, :3], full[:K, :3], rtol=1e-13)
    for k in (0, 7, 31):
        lw = oracle.unnormalized_log_weights(ll, lp, lq, 0.0, betas[k])
        assert full[k, 0] == lw.max()
        assert full[k, 1] == pytest.approx(np.exp(lw - lw.max()).sum(), rel=1e-12)


def test_split_max_sums_equal_fused(eng):
    x, ll, lp, lq = synth(50000, 4, 6)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    betas = [0.1, 0.7]
    fused = eng.weights_stats(lld, lpd, lqd, 0.0, betas)
    m, nn = eng.weights_max(lld, lpd, lqd, 0.0, betas)
    s = eng.weights_sums(lld, lpd, lqd, 0.0, betas, m)
    assert nn == 0 and np.array_equal(m, fused[:, 0]) and np.array_equal(s, fused[:, 1:3])


def test_nan_guard_and_inf_semantics(eng):
    from aspire_amd.samples import SMCSamples

    ll = np.array([0.0, np.nan, 1.0])
    s = SMCSamples(x=np.zeros((3, 1)), log_likelihood=ll, log_prior=np.zeros(3), log_q=np.zeros(3), beta=0.0, engine=eng)
    with pytest.raises(ValueError, match="Log weights contain NaN"):
        s.log_weights(0.5)
    # -inf likelihood rows get zero weight, not NaN (H6)
    ll = np.array([0.0, -np.inf, 1.0, 2.0])
    s = SMCSamples(x=np.zeros((4, 1)), log_likelihood=ll, log_prior=np.zeros(4), log_q=np.zeros(4), beta=0.0, engine=eng)
    assert math.isfinite(s.log_evidence_ratio(0.5))
    assert eng.count_nonfinite(eng.asarray(ll)) == (0, 1)


# ---- K3: beta bisection -----------------------------------------------------------------------
def test_determine_beta_matches_reference_golden(eng, golden):
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.samples import SMCSamples
    from aspire_amd.targets import DiagGaussianMixture

    g = golden["ref_beta"]
    for n, d, seed, b0, tol, ti, b_ref in g["cases"]:
        n, d, seed, ti = int(n), int(d), int(seed), int(ti)
        x, ll, lp, lq = synth(n, d, 0, 2.0) if seed == 0 else synth(n, d, seed)
        lik = DiagGaussianMixture.isotropic(d, normalized=False)
        sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, engine=eng), xp=np, engine=eng)
        sp.adaptive, sp.adaptive_min_beta_step = True, False
        sp.target_efficiency = [0.5, (0.3, 0.9)][ti]
        sp.target_efficiency_rate = 1.0
        s = SMCSamples(x=eng.asarray(x), log_likelihood=eng.asarray(ll), log_prior=eng.asarray(lp), log_q=eng.asarray(lq),
                       beta=b0, engine=eng)
        b, _ = sp.determine_beta(s, b0, np.nan, 0.0, max_beta_step=1.0, beta_tolerance=tol)
        assert b == b_ref, (n, b0, tol, ti, b, b_ref)


def test_device_side_bisection_matches_reference_golden(eng, golden):
    """asmc_find_beta (whole search on device, no host round trips) returns the reference's beta* bit-for-bit."""
    g = golden["ref_beta"]
    from aspire_amd import smc_math

    for n, d, seed, b0, tol, ti, b_ref in g["cases"]:
        n, d, seed, ti = int(n), int(d), int(seed), int(ti)
        x, ll, lp, lq = synth(n, d, 0, 2.0) if seed == 0 else synth(n, d, seed)
        target = smc_math.current_target_efficiency([0.5, (0.3, 0.9)][ti], 1.0, b0)
        b, eff1, conv, passes, n_nan, trip, trip_one = eng.find_beta(*dev(eng, ll, lp, lq), b0, target, tol)
        assert conv and n_nan == 0 and b == b_ref, (n, b0, tol, ti, b, b_ref)
        assert passes <= 9
        # the triples it hands back are the log-sum-exp reductions at beta* and at 1 (shifted by the closed-form
        # maximum rather than the searched one: same ESS and evidence ratio to rounding)
        n_g = ll.size
        want_b, want_1 = smc_math.global_stats(eng, Comm(), *dev(eng, ll, lp, lq), b0, [b, 1.0], n_g)
        got_1 = smc_math.Stats(*trip_one, n_g)
        assert got_1.m == want_1.m and smc_math.ess(got_1) == pytest.approx(smc_math.ess(want_1), rel=1e-12)
        assert eff1 == pytest.approx(smc_math.ess(want_1) / n_g, rel=1e-12)
        if trip is not None:
            got_b = smc_math.Stats(*trip, n_g)
            assert got_b.m == pytest.approx(want_b.m, rel=1e-12, abs=1e-12)
            assert smc_math.ess(got_b) == pytest.approx(smc_math.ess(want_b), rel=1e-12)
            assert smc_math.log_evidence_ratio(got_b) == pytest.approx(smc_math.log_evidence_ratio(want_b), rel=1e-12, abs=1e-13)
        else:
            assert b == b0
    # ESS(1.0) >= target short-circuits to beta* = 1 (smc/base.py:170-175)
    z = eng.asarray(np.zeros(64))
    assert eng.find_beta(z, z, z, 0.0, 0.5, 1e-6)[0] == 1.0
    # NaN inputs are reported
    bad = eng.asarray(np.array([0.0, np.nan, 1.0, 2.0]))
    assert eng.find_beta(bad, z[:4].contiguous(), z[:4].contiguous(), 0.0, 0.5, 1e-6)[4] > 0


# ---- K6: PCG64, cdf, search ---------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 1000, 65536, 65537, 300001])
def test_pcg64_uniforms_bit_exact_vs_numpy(eng, n):
    from aspire_amd.smc_math import pcg64_state

    for seed, offset in ((0, 0), (123, 5), (9, 70000)):
        rng = np.random.default_rng(seed)
        st = pcg64_state(rng)
        u = eng.uniforms_pcg64(st, offset, n).cpu().numpy()
        ref = np.random.default_rng(seed).random(offset + n)[offset:]
        assert np.array_equal(u, ref)


def _weights(n, seed, kind):
    g = np.random.default_rng(seed)
    if kind == "smooth":
        w = np.exp(g.normal(size=n))
    elif kind == "heavy":  # spans hundreds of binades, exact zeros, a dominant weight
        w = np.exp(60 * g.normal(size=n))
        w[g.integers(0, n, n // 10)] = 0.0
    elif kind == "tiny_first":
        w = np.exp(g.normal(size=n))
        k = min(5, n)
        w[:k] = [0.0, 5e-324, 1e-310, 3e-300, 1e-200][:k]
    elif kind == "equal":
        w = np.full(n, 1.0 / n)
    elif kind == "ties":  # dyadic weights: every add is a round-to-even tie candidate
        w = np.ldexp(1.0, -g.integers(1, 60, n).astype(np.int64)).astype(np.float64)
    if kind == "tiny_first" or w.sum() == 0:
        return w
    return w / w.sum()


@pytest.mark.parametrize("kind", ["smooth", "heavy", "tiny_first", "equal", "ties"])
@pytest.mark.parametrize("n", [1, 5, 4096, 4097, 100003, 1 << 20])
def test_exact_cdf_is_numpy_cumsum_bitwise(eng, n, kind):
    w = _weights(n, 17 + n, kind)
    cdf, total = eng.cdf(eng.asarray(w), "exact", 0.0)
    ref = np.cumsum(w)
    got = cdf.cpu().numpy()
    assert np.array_equal(got, ref), np.flatnonzero(got != ref)[:5]
    assert total == ref[-1]


@pytest.mark.parametrize("kind", ["growth", "steps", "late_steps", "huge", "zeros_then_growth"])
@pytest.mark.parametrize("n", [100, 2048, 6000, 300001])
def test_exact_cdf_tiles_with_many_binade_crossings(eng, n, kind):
    """Tiles in which the running sum leaves SEVERAL binades (exact_tile_fast: predicted crossings verified one by one; more than
    64 of them, or a failed verification, fall back to one block-wide pass per crossing): still numpy's cumsum bit for bit."""
    g = np.random.default_rng(n + len(kind))
    if kind == "growth":  # every one of the first 200 adds at least doubles the sum (more crossings than the fast path takes)
        w = np.exp(g.normal(size=n))
        k = min(200, n)
        w[:k] = np.ldexp(1.0 + g.random(k), 3 * np.arange(k) - 700)
    elif kind == "steps":  # the scale jumps by 2^7 every 300 elements: several crossings inside most early tiles
        w = np.exp(g.normal(size=n)) * np.ldexp(1.0, np.minimum(7 * (np.arange(n) // 300), 1100) - 200)
    elif kind == "late_steps":  # smooth first, then three jumps inside one late tile
        w = np.exp(g.normal(size=n))
        for j, at in enumerate((n // 2, n // 2 + 37, n // 2 + 300)):
            w[at:] *= 2.0 ** (11 + j)
    elif kind == "huge":  # sums near the top of the range
        w = np.exp(g.normal(size=n)) * 1e300 / n
    else:  # zeros in front, then growth from subnormals
        w = np.zeros(n)
        k = n // 3
        w[k:] = np.ldexp(1.0 + g.random(n - k), np.minimum(4 * np.arange(n - k) - 1074, 0))
    cdf, total = eng.cdf(eng.asarray(w), "exact", 0.0)
    ref = np.cumsum(w)
    got = cdf.cpu().numpy()
    assert np.array_equal(got, ref), np.flatnonzero(got != ref)[:5]
    assert total == ref[-1]
    # and behind an exact carry
    cdf2, total2 = eng.cdf(eng.asarray(w), "exact", 0.37)
    ref2 = np.cumsum(np.concatenate([[0.37], w]))[1:]
    assert np.array_equal(cdf2.cpu().numpy(), ref2) and total2 == ref2[-1]


def test_exact_cdf_fuzz_against_numpy():
    """tools/fuzz_exact_cdf.py, a short run: random lengths and weight laws with binade crossings everywhere, with and without a
    carry - numpy's cumsum bit for bit (20 000 rounds over four seeds ran clean on the round-5 tree)."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_exact_cdf.py")
    out = subprocess.run([sys.executable, tool], env=dict(os.environ, ROUNDS="150", SEED="7"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("kind", ["smooth", "heavy", "equal"])
@pytest.mark.parametrize("n", [1_100_003, 8_000_000])
def test_exact_cdf_bitwise_beyond_one_chain_chunk(kind, n):
    """Config 4's replicated scan covers all 8M particles of the node: > 512 tiles, so the chain kernel restages its
    LDS tile records several times.  Still bit-identical to numpy's sequential cumsum."""
    from aspire_amd.engine import HipEngine

    big = HipEngine(0, n_max=n, d_max=1)
    w = _weights(n, 29, kind)
    cdf, total = big.cdf(big.asarray(w), "exact", 0.0)
    ref = np.cumsum(w)
    got = cdf.cpu().numpy()
    assert np.array_equal(got, ref), np.flatnonzero(got != ref)[:5]
    assert total == ref[-1]
    # resampling at that size: indices equal numpy's Generator.choice for the same generator
    if kind == "smooth":
        p = w / w.sum()
        want = np.random.default_rng(5).choice(n, size=100000, replace=True, p=p)
        cdfn = big.cdf_normalize_last(big.cdf(big.asarray(p), "exact", 0.0, want_total=False)[0])
        u = big.asarray(np.random.default_rng(5).random(100000))
        assert np.array_equal(big.search(cdfn, u).cpu().numpy(), want)
    big.close()


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("n", [7, 4097, 300001])
def test_cdf_fused_normalisation_is_divide_by_last(eng, n, mode):
    """normalize=True == numpy's `cdf /= cdf[-1]` applied to the same scan (bitwise), last element exactly 1."""
    w = eng.asarray(_weights(n, 3 + n, "smooth") * 0.37)
    plain, total = eng.cdf(w, mode, 0.0)
    fused, total2 = eng.cdf(w, mode, 0.0, normalize=True)
    assert total2 == total
    # (numpy division: torch divides a tensor by a Python scalar through a reciprocal multiply)
    assert np.array_equal(fused.cpu().numpy(), plain.cpu().numpy() / total) and float(fused[-1]) == 1.0


def test_exact_cdf_with_carry_chains_like_one_array(eng):
    w = _weights(50000, 3, "smooth")
    a, ta = eng.cdf(eng.asarray(w[:20000]), "exact", 0.0)
    b, tb = eng.cdf(eng.asarray(w[20000:]), "exact", ta)
    ref = np.cumsum(w)
    assert np.array_equal(np.concatenate([a.cpu().numpy(), b.cpu().numpy()]), ref) and tb == ref[-1]


def test_fast_cdf_close_and_monotone(eng):
    w = _weights(300001, 4, "smooth")
    cdf, total = eng.cdf(eng.asarray(w), "fast", 0.0)
    got = cdf.cpu().numpy()
    np.testing.assert_allclose(got, np.cumsum(w), rtol=1e-12)
    assert np.all(np.diff(got) >= 0) and total == got[-1]


def test_search_is_searchsorted_right(eng):
    g = np.random.default_rng(5)
    cdf = np.cumsum(g.random(100000))
    cdf /= cdf[-1]
    u = np.concatenate([g.random(50000), cdf[::997][:50], [0.0]])  # includes exact hits on cdf values
    u = u[u < 1.0]
    idx = eng.search(eng.asarray(cdf), eng.asarray(u)).cpu().numpy()
    assert np.array_equal(idx, np.searchsorted(cdf, u, side="right"))


def test_resample_indices_match_reference_golden(eng, golden):
    from aspire_amd.samples import SMCSamples

    g = golden["ref_resample"]
    for n, d, seed, b0, b, n_out in g["cases"]:
        n, d, seed, n_out = int(n), int(d), int(seed), int(n_out)
        x, ll, lp, lq = synth(n, d, seed)
        rng = np.random.default_rng(1000 + seed)
        s = SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=b0, engine=eng)
        out = s.resample(b, n_samples=n_out, rng=rng)
        idx = g[f"n{n}_b{b0}_t{b}_o{n_out}_idx"]
        assert np.array_equal(out.x, x[idx]), (n, b0, b, n_out)
        assert np.array_equal(out.log_likelihood, ll[idx]) and np.array_equal(out.log_q, lq[idx])
        assert np.array_equal(rng.random(3), g[f"n{n}_b{b0}_t{b}_o{n_out}_next_u"])
    x, ll, lp, lq = synth(2000, 4, 32)
    s = SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=0.4, engine=eng)
    out = s.resample(0.4, n_samples=50, rng=np.random.default_rng(77))
    assert np.array_equal(out.x, x[g["samebeta_idx"]])


def test_resample_indices_vs_oracle_1m(eng, oracle):
    """Full size: 1M particles, exact mode == oracle (sequential cumsum) indices."""
    from aspire_amd import smc_math
    from aspire_amd.comm import Comm

    n = 1 << 20
    x, ll, lp, lq = synth(n, 2, 99)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    idx, _ = smc_math.resample_indices(eng, Comm(), lld, lpd, lqd, 0.0, 0.02, n, np.random.default_rng(5), mode="exact")
    ref = oracle.resample_indices(ll, lp, lq, 0.0, 0.02, np.random.default_rng(5).random(n))
    assert np.array_equal(idx.cpu().numpy(), ref)
    idx_f, _ = smc_math.resample_indices(eng, Comm(), lld, lpd, lqd, 0.0, 0.02, n, np.random.default_rng(5), mode="fast")
    diff = idx_f.cpu().numpy() != ref
    assert diff.sum() <= 4 and np.all(np.abs(idx_f.cpu().numpy()[diff] - ref[diff]) == 1)


# ---- K7 gather, K11 compaction ----------------------------------------------------------------------
@pytest.mark.parametrize("d,dtype", [(32, torch.float64), (32, torch.float32), (4, torch.float64), (3, torch.float64),
                                     (5, torch.float32), (128, torch.float64)])
def test_gather_rows_exact(eng, d, dtype):
    g = np.random.default_rng(7)
    n, n_out = 20011, 30007
    x = torch.as_tensor(g.normal(size=(n, d))).to(dtype)
    ll, lp, lq = (g.normal(size=n) for _ in range(3))
    idx = g.integers(0, n, n_out)
    xo, a, b, c = eng.gather(eng.asarray(idx, dtype=torch.int64), x.to(eng.device), *dev(eng, ll, lp, lq))
    assert torch.equal(xo.cpu(), x[idx]) and np.array_equal(a.cpu().numpy(), ll[idx])
    assert np.array_equal(b.cpu().numpy(), lp[idx]) and np.array_equal(c.cpu().numpy(), lq[idx])


def test_compact_valid_copies_nothing_when_every_row_is_valid(eng):
    g = np.random.default_rng(8)
    n, d = 5000, 5
    x, ll, lp, lq = dev(eng, g.normal(size=(n, d)), g.normal(size=n), g.normal(size=n), g.normal(size=n))
    got = eng.compact_valid(x, ll, lp, lq)
    assert all(a is b for a, b in zip(got, (x, ll, lp, lq)))


def test_compact_valid_matches_oracle(eng, oracle):
    g = np.random.default_rng(8)
    n, d = 10007, 5
    x = g.normal(size=(n, d))
    ll, lp, lq = g.normal(size=n), g.normal(size=n), g.normal(size=n)
    ll[g.integers(0, n, 500)] = -np.inf
    lp[g.integers(0, n, 500)] = np.inf
    ll[g.integers(0, n, 100)] = np.nan
    got = eng.compact_valid(*dev(eng, x, ll, lp, lq))
    ref = oracle.compact_valid(x, ll, lp, lq)
    for a, b in zip(got, ref):
        assert np.array_equal(a.cpu().numpy(), b)


# ---- proposal draw, densities, moments -----------------------------------------------------------------
def test_philox_normals_match_oracle_and_are_normal(eng, oracle):
    d, n = 6, 4096
    mu, sigma = np.linspace(-1, 1, d), np.linspace(0.5, 2, d)
    x, lq = eng.gaussian_draw(n, d, torch.float64, eng.asarray(mu), eng.asarray(sigma), 1234, 10, 3)
    xn = x.cpu().numpy()
    for i in (0, 1, 4095):
        xi, _ = oracle.pcn_noise(1234, 10 + i, 3, d)
        np.testing.assert_allclose(xn[i], mu + sigma * xi, rtol=1e-12, atol=1e-13)
    z = (xn - mu) / sigma
    ref_lq = -0.5 * (z * z).sum(1) - np.log(sigma).sum() - 0.5 * d * np.log(2 * np.pi)
    np.testing.assert_allclose(lq.cpu().numpy(), ref_lq, rtol=1e-12)
    from scipy import stats

    big, _ = eng.gaussian_draw(200000, 4, torch.float64, eng.asarray(np.zeros(4)), eng.asarray(np.ones(4)), 7, 0, 0, want_lq=False)
    assert stats.kstest(big.cpu().numpy().ravel(), "norm").pvalue > 1e-3


def test_box_muller_functions_agree_with_libm_to_ulps(eng, oracle):
    """The kernels' own fp64 log / sqrt / sin / cos of the Box-Muller transform (restricted domains, asmc_pcn_dev.h)
    against the oracle's libm on the same Philox words: 65 536 normals, tails included."""
    d, n = 32, 2048
    x, _ = eng.gaussian_draw(n, d, torch.float64, eng.asarray(np.zeros(d)), eng.asarray(np.ones(d)), 99, 5, 1, want_lq=False)
    xn = x.cpu().numpy()
    ref = np.stack([oracle.pcn_noise(99, 5 + i, 1, d)[0] for i in range(n)])
    err = np.abs(xn - ref)
    # the oracle rounds the angle 2 pi u before taking cos / sin (half an ulp of up to 2 pi = 4.4e-16 rad, times the pair's
    # radius); the kernels reduce u by quarter turns exactly.  Beyond that: a few ulp of the value itself.  (Against an
    # 80-bit evaluation the kernels are the closer of the two: tests/tools/bm_check.py, profiles/r02_box_muller_accuracy.txt.)
    radius = np.sqrt(ref[:, 0::2] ** 2 + ref[:, 1::2] ** 2).repeat(2, axis=1)
    tol = 8e-16 * radius + 4 * np.spacing(np.abs(ref)) + 2e-16
    assert np.all(err <= tol), float((err / tol).max())
    assert np.abs(ref).max() > 3.5  # the sample reaches into the tails


@pytest.mark.parametrize("d,C,dtype", [(32, 1, torch.float64), (4, 2, torch.float64), (7, 3, torch.float32), (128, 2, torch.float64)])
def test_mixture_logpdf_vs_oracle(eng, oracle, d, C, dtype):
    g = np.random.default_rng(9)
    n = 5003
    mu, var = g.normal(size=(C, d)), g.uniform(0.5, 2.0, size=(C, d))
    logw = np.log(np.full(C, 1.0 / C)) - 0.5 * d * np.log(2 * np.pi) - 0.5 * np.log(var).sum(1)
    x = torch.as_tensor(g.normal(size=(n, d)) * 2).to(dtype)
    got = eng.mixture_logpdf(x.to(eng.device), eng.make_mixture(logw, mu, 1 / var)).cpu().numpy()
    ref = oracle.Mixture(logw, mu, 1 / var).logpdf(x.double().numpy())
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("d", [2, 3, 4, 12, 32, 48, 64, 100, 128])
def test_moments_vs_oracle(eng, oracle, d):
    g = np.random.default_rng(10)
    n = 30011
    x = g.normal(size=(n, d)) @ g.normal(size=(d, d)) + 3.0
    xd = eng.asarray(x)
    mean = eng.colsum(xd) / n
    cov = eng.centered_gram(xd, mean) / (n - 1)
    rm, rc = oracle.moments(x)
    np.testing.assert_allclose(mean, rm, rtol=1e-12)
    np.testing.assert_allclose(cov, rc, rtol=1e-10, atol=1e-12)


# ---- K8/K9: pCN ------------------------------------------------------------------------------------------
def _pcn_setup(eng, n, d, seed, dtype=torch.float64, C=1):
    g = np.random.default_rng(seed)
    x = 1.5 * g.normal(size=(n, d))
    A = g.normal(size=(d, d)) / np.sqrt(d)
    cov = A @ A.T + 0.5 * np.eye(d)
    L = np.linalg.cholesky(cov)
    Linv = np.linalg.inv(L)
    mu = 0.1 * g.normal(size=d)
    mixes = []
    for k in range(3):
        m = g.normal(size=(C, d)) * (0.5 if k < 2 else 0.0)
        v = g.uniform(0.7, 1.5, size=(C, d)) * (2.25 if k == 2 else 1.0)
        lw = np.log(np.full(C, 1.0 / C)) - 0.5 * d * np.log(2 * np.pi) - 0.5 * np.log(v).sum(1)
        mixes.append((lw, m, 1 / v))
    return x, mu, np.tril(L), np.tril(Linv), mixes


@pytest.mark.parametrize("d,C", [(32, 1), (4, 2), (7, 1), (2, 1), (20, 1), (20, 3), (48, 1), (100, 2), (3, 1), (127, 1)])
def test_pcn_step_vs_oracle(eng, oracle, d, C):
    """One fused pCN step == the oracle's restatement of the same specification: proposals to 1e-12,
    accept decisions identical except where log u is within 1e-9 of log alpha.  Every d <= 128 runs on the
    register-resident / matrix-core kernels - a d without kernels of its own zero-padded to the next width
    (asmc_pcn_mutate) - never on the generic LDS kernel."""
    n = 3000
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 21 + d, C=C)
    om = [oracle.Mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    dm = [eng.make_mixture(*m) for m in mixes]
    xd, lld, lpd, lqd = dev(eng, x, ll, lp, lq)
    rho, beta, seed, gid0, step = 0.4, 0.37, 4242, 1000, 5
    eng.profile(True)
    n_acc, rho_hist, rho_out = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv),
                                              dm[0], dm[1], dm[2], seed, gid0, rho, 1, step, 0.234, False)
    rep = eng.profile_report()
    eng.profile(False)
    assert not any(k.startswith(("k_pcn_step_generic", "k_pcn_propose")) for k in rep), sorted(rep)
    assert ("k_pad_rows" in rep) == (d not in (4, 8, 16, 32, 64, 128))
    xr, llr, lpr, lqr = x.copy(), ll.copy(), lp.copy(), lq.copy()
    acc_ref = oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, om[0], om[1], om[2], seed, gid0, step)
    got = xd.cpu().numpy()
    # (the matrix-core kernels always step on the whitened state: a rejected row comes back through y -> x with rounding)
    moved_g = np.any(np.abs(got - x) > 1e-9 * (1 + np.abs(x)), axis=1)
    moved_r = np.any(xr != x, axis=1)
    disagree = moved_g != moved_r
    assert disagree.sum() <= 2, disagree.sum()  # only razor-edge accept decisions may differ
    same = ~disagree
    np.testing.assert_allclose(got[same], xr[same], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(lld.cpu().numpy()[same], llr[same], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lqd.cpu().numpy()[same], lqr[same], rtol=1e-11, atol=1e-11)
    assert abs(int(n_acc[0]) - acc_ref) <= 2 and rho_out == rho and rho_hist[0] == rho
    assert 0.02 < n_acc[0] / n < 0.98


@pytest.mark.parametrize("d,step_fn,dtype", [(20, "pcn", torch.float64), (7, "tpcn", torch.float64), (48, "pcn", torch.float64),
                                             (100, "tpcn", torch.float64), (20, "pcn", torch.float32)])
def test_padded_dimensions_multi_step_vs_oracle(eng, oracle, d, step_fn, dtype):
    """Four steps of a problem whose d has no kernels of its own (zero-padded to 8 / 32 / 64 / 128 inside asmc_pcn_mutate):
    pCN and tpCN (the Student-t correction and the Gamma shape keep the REAL dimension) against the oracle at the real d."""
    n, nu = 3000, 6.5
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 77 + d)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    xd = torch.as_tensor(x).to(dtype).to(eng.device)
    xr = xd.double().cpu().numpy().copy()
    ll, lp, lq = (m.logpdf(xr) for m in om)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    rho, beta, seed = 0.3, 0.55, 991
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0], dm[1], dm[2],
                                 seed, 70, rho, 4, 20, 0.234, False, "f64", nu if step_fn == "tpcn" else 0.0)
    rep = eng.profile_report()
    eng.profile(False)
    assert "k_pad_rows" in rep and "k_unpad_rows" in rep and not any(k.startswith("k_pcn_step_generic") for k in rep), sorted(rep)
    llr, lpr, lqr = ll.copy(), lp.copy(), lq.copy()
    acc_ref = []
    for t in range(4):
        if step_fn == "tpcn":
            acc_ref.append(oracle.tpcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, om[0], om[1], om[2], seed, 70, 20 + t))
        else:
            acc_ref.append(oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, om[0], om[1], om[2], seed, 70, 20 + t))
    got = xd.double().cpu().numpy()
    tol = 1e-9 if dtype == torch.float64 else 2e-5
    edge = 3 if dtype == torch.float64 else 40
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    assert (~close).sum() <= edge, (~close).sum()
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= edge) and 0.02 < np.mean(n_acc) / n < 0.98
    if dtype == torch.float64:
        np.testing.assert_allclose(lld.cpu().numpy()[close], llr[close], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("d,dtype", [(32, torch.float64), (8, torch.float64), (32, torch.float32)])
def test_pcn_whitened_state_multi_step_vs_oracle(eng, oracle, d, dtype):
    """n_steps >= 4 with plain Gaussian targets runs on the whitened state (x -> y once, one mat-vec per step,
    y -> x at the end).  Against the oracle's x-space restatement over 4 steps: same accept decisions up to
    razor-edge cases, positions to 1e-9 (fp64 storage) / 1e-5 (fp32 storage), carried log-probs consistent."""
    n = 3000
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 31 + d)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    xd = torch.as_tensor(x).to(dtype).to(eng.device)
    xr = xd.double().cpu().numpy().copy()
    x0 = xr.copy()
    ll, lp, lq = (m.logpdf(xr) for m in om)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    rho, beta, seed = 0.35, 0.6, 777
    n_acc, rho_hist, _ = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0],
                                        dm[1], dm[2], seed, 50, rho, 4, 10, 0.234, False)
    llr, lpr, lqr = ll.copy(), lp.copy(), lq.copy()
    acc_ref = [oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, om[0], om[1], om[2], seed, 50, 10 + t) for t in range(4)]
    got = xd.double().cpu().numpy()
    tol = 1e-9 if dtype == torch.float64 else 2e-5
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    assert (~close).sum() <= (3 if dtype == torch.float64 else 40), (~close).sum()  # razor-edge decisions (fp32 rounding of y widens the edge)
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= (3 if dtype == torch.float64 else 40))
    # carried log-probabilities equal the densities at the stored positions
    np.testing.assert_allclose(lld.cpu().numpy(), om[0].logpdf(got), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(lqd.cpu().numpy(), om[2].logpdf(got), rtol=1e-11, atol=1e-11)
    # particles that never moved come back within rounding of where they started
    still = np.all(np.abs(xr - x0) == 0, axis=1) & close
    assert np.all(np.abs(got[still] - x0[still]) <= (1e-13 if dtype == torch.float64 else 1e-5) * (1 + np.abs(x0[still])))


def test_pcn_fast_noise_vs_oracle(eng, oracle):
    """ASMC_NOISE_F32 (hardware fp32 Box-Muller): proposals within 2e-6 of the libm restatement, accept
    decisions equal except for a handful of razor-edge cases; noise is standard normal."""
    d, n = 32, 4000
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 5)
    om = [oracle.Mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    dm = [eng.make_mixture(*m) for m in mixes]
    xd, lld, lpd, lqd = dev(eng, x, ll, lp, lq)
    n_acc, _, _ = eng.pcn_mutate(xd, lld, lpd, lqd, 0.37, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0], dm[1],
                                 dm[2], 99, 7, 0.4, 1, 3, 0.234, False, "f32")
    xr, llr, lpr, lqr = x.copy(), ll.copy(), lp.copy(), lq.copy()
    acc_ref = oracle.pcn_step(xr, llr, lpr, lqr, 0.37, mu, L, Linv, 0.4, om[0], om[1], om[2], 99, 7, 3, noise="f32")
    got = xd.cpu().numpy()
    moved_g, moved_r = np.any(got != x, axis=1), np.any(xr != x, axis=1)
    assert (moved_g != moved_r).sum() <= 4
    same = moved_g == moved_r
    np.testing.assert_allclose(got[same], xr[same], rtol=1e-5, atol=2e-5)
    assert abs(int(n_acc[0]) - acc_ref) <= 4
    # distribution of the fast normals: recover xi from a pure-noise step (rho = 1, identity reference)
    from scipy import stats

    n2, d2 = 200000, 4
    tgt = eng.make_mixture([0.0], np.zeros((1, d2)), np.full((1, d2), 1e-30))  # flat target: always accept
    x0 = eng.asarray(np.zeros((n2, d2)))
    z = torch.zeros(n2, dtype=torch.float64, device=eng.device)
    eye = eng.asarray(np.eye(d2))
    na, _, _ = eng.pcn_mutate(x0, z.clone(), z.clone(), z.clone(), 1.0, eng.asarray(np.zeros(d2)), eye, eye, tgt, tgt, tgt,
                              5, 0, 1.0, 1, 0, 0.234, False, "f32")
    xi = x0.cpu().numpy()  # flat target, rho = 1, y = 0: every proposal is accepted and equals the noise draw
    assert int(na[0]) == n2
    assert stats.kstest(xi.ravel(), "norm").pvalue > 1e-3
    assert abs(np.mean(xi[:, 0] * xi[:, 1])) < 0.01 and abs(np.mean(xi[:, 2] * xi[:, 3])) < 0.01


def test_pcn_split_path_equals_fused(eng, oracle):
    n, d = 2000, 8
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 77)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    a = dev(eng, x, ll, lp, lq)
    b = dev(eng, x, ll, lp, lq)
    mud, Ld, Lid = dev(eng, mu, L, Linv)
    n_acc, _, _ = eng.pcn_mutate(*a, 0.6, mud, Ld, Lid, dm[0], dm[1], dm[2], 5, 0, 0.3, 1, 9, 0.234, False)
    xp, q0, q1 = eng.pcn_propose(b[0], mud, Ld, Lid, 0.3, 5, 0, 9)
    lln, lpn, lqn = (eng.mixture_logpdf(xp, m) for m in dm)
    nacc = eng.pcn_accept(b[0], xp, b[1], b[2], b[3], lln, lpn, lqn, q0, q1, 0.6, 5, 0, 9)
    # same proposals bit for bit (the proposal half IS the fused kernel); the densities of the split path come from the flat
    # mixture kernel, which sums the quadratic form in butterfly order: log-probabilities agree to rounding, so an accept
    # decision can flip only on a razor edge
    assert abs(nacc - int(n_acc[0])) <= 1
    moved_a, moved_b = (a[0] != dev(eng, x)[0]).any(dim=1), (b[0] != dev(eng, x)[0]).any(dim=1)
    same = moved_a == moved_b
    assert int((~same).sum()) <= 1
    assert torch.equal(a[0][same], b[0][same])
    for u, v in zip(a[1:], b[1:]):
        torch.testing.assert_close(u[same], v[same], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("d,nu,dtype,noise", [(8, 0.0, torch.float64, "f64"), (32, 0.0, torch.float64, "f64"),
                                              (4, 5.0, torch.float64, "f64"), (16, 7.5, torch.float64, "f64"),
                                              (32, 0.0, torch.float32, "f64"), (32, 0.0, torch.float64, "f32"),
                                              (8, 6.0, torch.float64, "f32")])
def test_pcn_ysplit_session_equals_fused(eng, oracle, d, nu, dtype, noise):
    """Whitened-state split session (propose -> caller's densities -> accept, state coordinate-major in the ctx) == the fused
    whitened-state step loop on the same counters: same proposals bit for bit, decisions identical off a razor edge."""
    n, steps = 3000, 3
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 300 + d)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    a = [eng.asarray(x).to(dtype)] + list(dev(eng, ll, lp, lq))
    b = [eng.asarray(x).to(dtype)] + list(dev(eng, ll, lp, lq))
    mud, Ld, Lid = dev(eng, mu, L, Linv)
    n_acc, _, _ = eng.pcn_mutate(*a, 0.6, mud, Ld, Lid, dm[0], dm[1], dm[2], 5, 17, 0.3, steps, 4, 0.234, False, noise, nu)
    sess = eng.pcn_ysplit_begin(b[0], 0.6, mud, Ld, Lid, 5, 17, 0.3, 0.234, False, nu, noise)
    assert sess is not None
    x_before = b[0].clone()
    for t in range(steps):
        xp = eng.pcn_ysplit_propose(sess, 4 + t)
        lln, lpn, lqn = (eng.mixture_logpdf(xp, m) for m in dm)
        eng.pcn_ysplit_accept(sess, 4 + t, b[1], b[2], b[3], lln, lpn, lqn, n, t)
    assert torch.equal(b[0], x_before)  # the session owns the state until it ends
    nb, hist, rho = eng.pcn_ysplit_end(sess, steps)
    assert rho == 0.3 and np.all(hist == 0.3)
    assert np.all(np.abs(nb - n_acc) <= 2)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    close = ((a[0] - b[0]).abs() <= tol * (1 + a[0].abs())).all(dim=1)
    assert int((~close).sum()) <= 3
    # (fp32 storage: the split form evaluates the densities at the stored, fp32-rounded x'; the fused loop at the fp64 registers)
    lt = 1e-11 if dtype == torch.float64 else 1e-5
    for u, v in zip(a[1:], b[1:]):
        torch.testing.assert_close(u[close], v[close], rtol=lt, atol=lt)


@pytest.mark.parametrize("d,nu,n,noise", [(4, 0.0, 400, "f64"), (8, 6.0, 400, "f64"), (16, 0.0, 1, "f64"), (32, 4.0, 65, "f64")])
def test_pcn_ysplit_session_vs_test_double(eng, oracle, d, nu, n, noise):
    """... and == the host restatement the CPU suite runs the sampler on (adaptation included; ragged last tile, n = 1).
    (The hardware-fp32 noise mode agrees with libm only to 2e-6 - test_pcn_fast_noise... - and is pinned through the fused
    loop above instead.)"""
    from oracle_engine import OracleEngine

    steps = 4
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 350 + d)
    om = [oracle.Mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    ref = OracleEngine()
    out = []
    for e in (eng, ref):
        st = [e.asarray(np.array(v)) for v in (x, ll, lp, lq)]
        sess = e.pcn_ysplit_begin(st[0], 0.7, e.asarray(mu), e.asarray(L), e.asarray(Linv), 9, 1 << 40, 0.4, 0.234, True, nu,
                                  noise)
        for t in range(steps):
            xp = e.pcn_ysplit_propose(sess, t)
            new = [e.asarray(m.logpdf(xp.cpu().numpy().astype(np.float64))) for m in om]
            e.pcn_ysplit_accept(sess, t, st[1], st[2], st[3], *new, n, t)
        n_acc, hist, rho = e.pcn_ysplit_end(sess, steps)
        out.append(([v.cpu().numpy() for v in st], n_acc, hist, rho))
    (sa, na, ha, ra), (sb, nb, hb, rb) = out
    np.testing.assert_array_equal(na, nb)
    np.testing.assert_allclose(ha, hb, rtol=1e-14)
    assert abs(ra - rb) < 1e-14
    for u, v in zip(sa, sb):
        np.testing.assert_allclose(u, v, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("d,nu", [(8, 0.0), (32, 5.0)])
def test_pcn_ysplit_accept_with_log_jacobian(eng, oracle, d, nu):
    """Chain in a preconditioned space: carried / proposed log-Jacobians join the tempered log-target and follow the accepted
    state - first step == the x-state propose / accept pair with the same arrays; several steps == the test double."""
    from oracle_engine import OracleEngine

    n = 3000
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 400 + d)
    om = [oracle.Mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    g = np.random.default_rng(5)
    lj0 = g.normal(size=n)
    jac = lambda xp: 0.3 * np.sin(xp).sum(1)  # noqa: E731  (any smooth function of the proposal stands in for log|J|)
    mud, Ld, Lid = dev(eng, mu, L, Linv)
    a = list(dev(eng, x, ll, lp, lq, lj0))
    b = list(dev(eng, x, ll, lp, lq, lj0))
    zp, q0, q1 = eng.pcn_propose(a[0], mud, Ld, Lid, 0.35, 11, 3, 0, nu=nu)
    new = [eng.asarray(m.logpdf(zp.cpu().numpy())) for m in om]
    na = eng.pcn_accept(a[0], zp, a[1], a[2], a[3], *new, q0, q1, 0.6, 11, 3, 0, logj_old=a[4],
                        logj_new=eng.asarray(jac(zp.cpu().numpy())))
    sess = eng.pcn_ysplit_begin(b[0], 0.6, mud, Ld, Lid, 11, 3, 0.35, 0.234, False, nu)
    zq = eng.pcn_ysplit_propose(sess, 0)
    torch.testing.assert_close(zq, zp, rtol=1e-13, atol=1e-13)
    new = [eng.asarray(m.logpdf(zq.cpu().numpy())) for m in om]
    eng.pcn_ysplit_accept(sess, 0, b[1], b[2], b[3], *new, n, 0, logj=b[4], logj_new=eng.asarray(jac(zq.cpu().numpy())))
    nb, _, _ = eng.pcn_ysplit_end(sess, 1)
    assert abs(int(nb[0]) - na) <= 1 and 0 < na < n
    same = (a[4] == b[4])
    assert int((~same).sum()) <= 1 and int((a[4] != eng.asarray(lj0)).sum()) >= na - 1  # the Jacobian moved with the state
    for u, v in zip(a, b):
        torch.testing.assert_close(u[same], v[same], rtol=1e-11, atol=1e-11)
    # several adapted steps against the host restatement
    out = []
    for e in (eng, OracleEngine()):
        st = [e.asarray(np.array(v)) for v in (x[:300], ll[:300], lp[:300], lq[:300], lj0[:300])]
        sess = e.pcn_ysplit_begin(st[0], 0.7, e.asarray(mu), e.asarray(L), e.asarray(Linv), 9, 5, 0.4, 0.234, True, nu)
        for t in range(3):
            xp = e.pcn_ysplit_propose(sess, t).cpu().numpy().astype(np.float64)
            e.pcn_ysplit_accept(sess, t, st[1], st[2], st[3], *[e.asarray(m.logpdf(xp)) for m in om], 300, t, logj=st[4],
                                logj_new=e.asarray(jac(xp)))
        n_acc, hist, rho = e.pcn_ysplit_end(sess, 3)
        out.append(([v.cpu().numpy() for v in st], n_acc, rho))
    np.testing.assert_array_equal(out[0][1], out[1][1])
    assert abs(out[0][2] - out[1][2]) < 1e-14
    for u, v in zip(out[0][0], out[1][0]):
        np.testing.assert_allclose(u, v, rtol=1e-10, atol=1e-10)


def test_pcn_ysplit_unsupported_dimension_and_errors(eng):
    from aspire_amd._lib import AsmcError

    x, mu, L, Linv, _ = _pcn_setup(eng, 100, 6, 3)
    xd, mud, Ld, Lid = dev(eng, x, mu, L, Linv)
    assert eng.pcn_ysplit_begin(xd, 0.5, mud, Ld, Lid, 1, 0, 0.3) is None
    x, mu, L, Linv, _ = _pcn_setup(eng, 100, 8, 3)
    xd, mud, Ld, Lid = dev(eng, x, mu, L, Linv)
    with pytest.raises(AsmcError):
        eng.pcn_ysplit_begin(xd, 0.5, mud, Ld, Lid, 1, 0, 1.5)
    with pytest.raises(AsmcError):
        eng.pcn_ysplit_begin(xd, 0.5, mud, Ld, Lid, 1, 0, 0.3, nu=0.5)


def test_pcn_leaves_gaussian_target_invariant(eng):
    """Stationarity: particles ~ N(0, 1/2 I) stay N(0, 1/2 I) under many pCN steps at beta=1
    (ll = lp = -|x|^2/2), with the reference Gaussian deliberately mis-specified."""
    n, d = 200000, 4
    g = np.random.default_rng(1)
    x = g.normal(size=(n, d)) * np.sqrt(0.5)
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
    xd = eng.asarray(x)
    ll = eng.mixture_logpdf(xd, tgt)
    lp = ll.clone()
    lq = eng.mixture_logpdf(xd, q)
    L = np.diag(np.full(d, 0.9))
    n_acc, rho_hist, rho = eng.pcn_mutate(xd, ll, lp, lq, 1.0, eng.asarray(np.full(d, 0.2)), eng.asarray(L),
                                          eng.asarray(np.linalg.inv(L)), tgt, tgt, q, 11, 0, 0.5, 40, 0, 0.234, True)
    xs = xd.cpu().numpy()
    assert np.all(np.abs(xs.mean(0)) < 0.01) and np.all(np.abs(xs.var(0) - 0.5) < 0.01)
    # adaptation drives the acceptance towards 0.234 until the step size hits its 0.99 ceiling
    assert rho == 0.99 or abs(n_acc[-10:].mean() / n - 0.234) < 0.05
    assert rho_hist[1] > rho_hist[0] and np.all(np.diff(rho_hist) >= 0)  # acceptance above target -> step size grows
    np.testing.assert_allclose(ll.cpu().numpy(), -0.5 * (xs**2).sum(1), rtol=1e-12)  # carried log-probs stay consistent


def test_pcn_adapt_device_equals_host_formula(eng):
    from aspire_amd.samplers.smc import pcn_adapt

    n, d = 5000, 4
    g = np.random.default_rng(2)
    x = g.normal(size=(n, d))
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    xd = eng.asarray(x)
    ll = eng.mixture_logpdf(xd, tgt)
    lp, lq = ll.clone(), ll.clone()
    eye = eng.asarray(np.eye(d))
    n_acc, rho_hist, rho = eng.pcn_mutate(xd, ll, lp, lq, 0.5, eng.asarray(np.zeros(d)), eye, eye, tgt, tgt, tgt, 3, 0,
                                          0.6, 6, 0, 0.234, True)
    r = 0.6
    for t in range(6):
        assert rho_hist[t] == pytest.approx(r, rel=1e-13)
        r = pcn_adapt(r, n_acc[t] / n, 0.234, t)
    assert rho == pytest.approx(r, rel=1e-13)


# ---- end to end -----------------------------------------------------------------------------------------------------
def test_sampler_loop_history_matches_reference_golden(eng, golden):
    """Same whole-loop golden as the CPU suite, now with the HIP engine underneath (stub mutate on host)."""
    from test_host_logic import NumpyGaussFlow, StubSMC, _log_like

    g = golden["ref_loop"]
    for name, kind, kwargs in [("identity_adaptive", "identity", dict(adaptive=True, target_efficiency=0.5, beta_tolerance=1e-6)),
                               ("rw_adaptive", "rw", dict(adaptive=True, target_efficiency=0.5, beta_tolerance=1e-6))]:
        sp = StubSMC(log_likelihood=_log_like, log_prior=_log_like, dims=4, prior_flow=NumpyGaussFlow(4, 2.0, 5), xp=np,
                     rng=np.random.default_rng(9), engine=eng)
        sp.kind = kind
        sp.sampler_kwargs = {}
        if kind == "rw":  # the host stub works on numpy views of device tensors
            orig = sp.mutate

            def mutate(particles, beta, n_steps=None, _orig=orig):
                cpu = particles.to_numpy()
                cpu.x = torch.from_numpy(cpu.x)
                out = _orig(cpu, beta)
                return sp._wrap(eng.asarray(out.x), eng.asarray(out.log_likelihood), eng.asarray(out.log_prior),
                                eng.asarray(out.log_q), beta)

            sp.mutate = mutate
        out = sp.sample(2000, store_sample_history=False, **kwargs)
        assert np.array_equal(np.array(sp.history.beta), g[name + "_beta"])
        np.testing.assert_allclose(sp.history.ess, g[name + "_ess"], rtol=1e-10)
        np.testing.assert_allclose(sp.history.log_norm_ratio, g[name + "_log_norm_ratio"], rtol=1e-11, atol=1e-13)
        assert np.array_equal(np.asarray(out.x.cpu() if torch.is_tensor(out.x) else out.x), g[name + "_x_final"])


def test_quickstart_api_config1(eng):
    """BASELINE config 1 (README quickstart shape): Aspire(...).fit(...).sample_posterior(sampler="smc")."""
    from aspire_amd import Aspire, Samples
    from aspire_amd.samples import set_default_engine

    set_default_engine(eng)

    def log_likelihood(samples):
        return -0.5 * np.sum(samples.x**2, axis=-1)

    def log_prior(samples):
        return -0.5 * np.sum(samples.x**2, axis=-1)

    rng = np.random.default_rng(0)
    init = Samples(rng.normal(size=(2000, 4)))
    aspire = Aspire(log_likelihood=log_likelihood, log_prior=log_prior, dims=4, parameters=[f"x{i}" for i in range(4)],
                    flow_backend="gaussian")
    aspire.fit(init)
    post, hist = aspire.sample_posterior(sampler="smc", n_samples=500, sampler_kwargs=dict(n_steps=20),
                                         return_history=True, rng=np.random.default_rng(1))
    assert isinstance(post, Samples) and post.x.shape == (500, 4) and isinstance(post.x, np.ndarray)
    assert post.parameters == [f"x{i}" for i in range(4)] and post.log_w is None
    assert hist.beta[-1] == 1.0 and len(hist.ess) == len(hist.beta) == len(hist.mcmc_acceptance)
    assert abs(float(post.log_evidence) - 2 * math.log(math.pi)) < 5 * float(post.log_evidence_error) + 0.15
    assert aspire.n_likelihood_evaluations > 0


def test_fused_sampler_1m_d32_evidence(eng):
    """BASELINE config 2/3 size: 1M particles, d=32, fused pCN; log Z within 1 sigma-ish of (d/2) log pi."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 32, 1 << 20
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, engine=eng, seed=1),
                xp=np, engine=eng, rng=np.random.default_rng(0))
    out = sp.sample(n, sampler_kwargs=dict(n_steps=8), store_sample_history=False)
    true = 0.5 * d * math.log(math.pi)
    err = float(out.log_evidence_error)
    assert abs(float(out.log_evidence) - true) < max(4 * err, 0.02), (float(out.log_evidence), true, err)
    assert sp.history.beta[-1] == 1.0


@pytest.mark.parametrize("step_fn", ["tpcn", "pcn"])
def test_config5_shape_mixture_d128_mfma_kernels(eng, step_fn):
    """BASELINE config 5 shape at reduced N: d=128 two-component Gaussian-mixture likelihood (the
    examples/smc_example.py construction), N(0,I) prior, q = N(0, 3^2 I); runs on the generic pCN kernel
    (d > 32) with the blocked Gram kernel; log Z is analytic (prior x normalised mixture => log Z = log of the
    mixture's convolution with the prior at 0 ... here checked against importance-free closed form)."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 128, 16384
    lik = DiagGaussianMixture(np.stack([2 * np.ones(d), -2 * np.ones(d)]), np.stack([0.5 * np.ones(d), np.ones(d)]))
    prior = DiagGaussianMixture.isotropic(d, 0.0, 1.0)
    sp = HipSMC(log_likelihood=lik, log_prior=prior, dims=d, prior_flow=GaussianFlow(d, sigma=3.0, engine=eng, seed=4),
                xp=np, engine=eng, rng=np.random.default_rng(1))
    eng.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=6, step_fn=step_fn), store_sample_history=False, max_n_steps=40)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_tpcn_mm_step" if step_fn == "tpcn" else "k_pcn_mm_step"][0] >= 6 and "k_pcn_step_generic" not in rep
    if step_fn == "pcn":  # Gaussian reference: population moments through the matrix-core Gram kernel
        assert rep["k_gram_mm"][0] >= 1
    else:  # Student-t reference: fitted to a subsample on the device, scale variates from their own kernel (several steps per launch)
        assert 1 <= rep["k_gamma_draw"][0] <= rep["k_tpcn_mm_step"][0] and len(sp.history.mcmc_nu) == len(sp.history.beta)
        assert "k_student_mstep" in rep  # the whole EM on the device (asmc_student_fit)
    # Z = 0.5 * N(2; 0, (1 + 0.5) I) + 0.5 * N(-2; 0, (1 + 1) I)   (Gaussian convolution), per-dim product
    def lg(mu, var):
        return -0.5 * d * np.log(2 * np.pi * var) - 0.5 * d * mu * mu / var
    true = np.logaddexp(np.log(0.5) + lg(2.0, 1.5), np.log(0.5) + lg(2.0, 2.0))
    assert sp.history.beta[-1] == 1.0
    assert np.isfinite(float(out.log_evidence))
    # high-dimensional, 16k particles, six mutation steps per temperature: a loose sanity band on an under-resolved run (the
    # kernels' dispatch is what this test is about; the 1M-particle runs of test_gpu_fullsize.py carry the 3-sigma claim)
    assert abs(float(out.log_evidence) - true) < 0.2 * abs(true)
    assert 0.02 < np.mean(sp.history.mcmc_acceptance) < 0.98


def test_coupling_flow_split_path_config3_shape(eng):
    """BASELINE config 3 shape at reduced N: PyTorch coupling-flow proposal (dense layers on hipBLASLt) in the
    split propose / flow.log_prob / accept path."""
    from aspire_amd.flows import CouplingFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 8, 4096
    g = np.random.default_rng(0)
    flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float64, seed=1234)
    flow.fit(1.4 * g.normal(size=(4000, d)), n_epochs=30)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(2))
    out = sp.sample(n, sampler_kwargs=dict(n_steps=6), store_sample_history=False)
    true = 0.5 * d * math.log(math.pi)
    assert sp.history.beta[-1] == 1.0
    assert abs(float(out.log_evidence) - true) < 6 * float(out.log_evidence_error) + 0.1, (float(out.log_evidence), true)


# ---- coupling-flow log-density on the fp32 MFMA -----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d,n_layers,hidden,n,xdt", [
    (32, 4, 64, 4099, "f64"),   # BASELINE config 3's flow, ragged last tile
    (32, 4, 64, 4099, "f32"),
    (32, 6, 64, 1000, "f64"),   # does not fit LDS: streamed layer by layer
    (32, 1, 32, 33, "f64"),
    (4, 2, 32, 257, "f64"),     # padded half (2 -> 16)
    (20, 3, 64, 64, "f32"),
    (64, 2, 128, 777, "f64"),   # H = 32
    (48, 3, 64, 1, "f64"),
])
def test_coupling_logprob_vs_oracle(eng, oracle, d, n_layers, hidden, n, xdt):
    """HIP fp32-MFMA flow kernel vs the fp32 C oracle on the same inputs.  Both evaluate in fp32 with different
    summation orders (MFMA k-order is permuted), so the tolerance is fp32 rounding of the result:
    |delta| <= 1e-5 |log q| + 3e-4."""
    from conftest import random_coupling_flow

    flow = random_coupling_flow(d, n_layers, hidden)
    x = np.random.default_rng(11).normal(size=(n, d)) * 1.3
    if xdt == "f32":
        x = x.astype(np.float32).astype(np.float64)
    ws, bs = flow.export_layers()
    want = oracle.coupling_logprob(x, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    dev = flow.device_coupling(eng)
    xt = eng.asarray(x, dtype=torch.float32 if xdt == "f32" else torch.float64)
    got = eng.to_numpy(eng.coupling_logprob(xt, dev))
    assert np.all(np.isfinite(got))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=3e-4)


@pytest.mark.gpu
def test_coupling_logprob_vs_torch_modules(eng):
    """The same flow evaluated by its torch modules on the GPU (fp64 copy of the parameters = ground truth)."""
    from conftest import random_coupling_flow

    flow = random_coupling_flow(32, 4, 64)
    flow64 = random_coupling_flow(32, 4, 64, dtype=torch.float64).to(eng.device)
    x = torch.randn((20000, 32), device=eng.device, dtype=torch.float64, generator=torch.Generator(eng.device).manual_seed(1))
    got = eng.coupling_logprob(x, flow.device_coupling(eng))
    want = flow64.log_prob(x)
    torch.testing.assert_close(got, want, rtol=2e-5, atol=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("xdt,d,nu", [("f64", 32, 0.0), ("f32", 32, 0.0), ("f64", 8, 0.0), ("f64", 20, 0.0),
                                        ("f64", 32, 5.0), ("f64", 20, 5.0)])
def test_pcn_mutate_flow_vs_split_calls(eng, xdt, d, nu):
    """asmc_pcn_mutate_flow (whitened-state register kernels around the MFMA flow kernel, all steps enqueued on
    the device; d = 20 is zero-padded into the one-kernel step, d = 8 likewise) against the same steps issued one ABI call at a time in x
    space (propose / coupling_logprob / mixture_logpdf / accept), each of which is checked against the oracle
    above.  The whitened state rounds differently (1e-13 relative in fp64, 1e-6 in fp32 storage), so accept
    decisions may differ for razor-edge cases; everything else must agree, and the carried log-probabilities
    must equal the densities at the returned positions.  With adaptation on, the device-side step-size history
    follows the host formula."""
    from conftest import random_coupling_flow
    from aspire_amd.samplers.smc import pcn_adapt

    n, n_steps, beta, rho = 5000, 6, 0.35, 0.4
    dt = torch.float64 if xdt == "f64" else torch.float32
    flow = random_coupling_flow(d, 4, 64)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(3)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g).to(dt)
    t_ll = eng.make_mixture([0.0, -0.3], np.stack([np.full(d, 0.5), np.full(d, -0.5)]), np.ones((2, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(0.1 * np.arange(d) / d)
    A = np.eye(d) + 0.05 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    L, Linv = eng.asarray(A), eng.asarray(np.linalg.inv(A))

    def init():
        x = x0.clone()
        return x, eng.mixture_logpdf(x, t_ll), eng.mixture_logpdf(x, t_lp), eng.coupling_logprob(x, dev)

    xa, lla, lpa, lqa = init()
    n_acc, rho_hist, rho_out = eng.pcn_mutate_flow(xa, lla, lpa, lqa, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho,
                                                   n_steps, 5, 0.234, False, "f64", nu)
    xb, llb, lpb, lqb = init()
    acc_b = []
    for t in range(n_steps):
        xp, q0, q1 = eng.pcn_propose(xb, mu, L, Linv, rho, 77, 1000, 5 + t, nu=nu)
        lqn = eng.coupling_logprob(xp, dev)
        acc_b.append(eng.pcn_accept(xb, xp, llb, lpb, lqb, eng.mixture_logpdf(xp, t_ll), eng.mixture_logpdf(xp, t_lp), lqn,
                                    q0, q1, beta, 77, 1000, 5 + t))
    assert 0 < sum(acc_b) < n * n_steps
    assert rho_out == rho and np.all(rho_hist == rho)
    tol = 1e-9 if xdt == "f64" else 2e-5
    close = ((xa.double() - xb.double()).abs() <= tol * (1 + xb.double().abs())).all(dim=1)
    edge = 5 if xdt == "f64" else 60
    assert int((~close).sum()) <= edge, int((~close).sum())
    assert np.all(np.abs(n_acc - np.array(acc_b)) <= edge)
    if d == 20:  # round 4: zero-padded into the one-kernel step (above: agreement to the whitened state's rounding); the generic
        # composition it replaced is still there (ASMC_PCN_NOPAD=1) and is the very same kernels as the split calls: bit-identical
        xc, llc, lpc, lqc = init()
        os.environ["ASMC_PCN_NOPAD"] = "1"
        try:
            n_acc_c, _, _ = eng.pcn_mutate_flow(xc, llc, lpc, lqc, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho, n_steps, 5, 0.234,
                                                False, "f64", nu)
        finally:
            del os.environ["ASMC_PCN_NOPAD"]
        assert n_acc_c.tolist() == acc_b
        for a, b in ((xc, xb), (llc, llb), (lpc, lpb), (lqc, lqb)):
            assert torch.equal(a, b)
    # carried log-probabilities are the densities at the returned positions
    torch.testing.assert_close(lla, eng.mixture_logpdf(xa, t_ll), rtol=1e-9 if xdt == "f64" else 1e-4, atol=1e-9 if xdt == "f64" else 2e-3)
    torch.testing.assert_close(lqa, eng.coupling_logprob(xa, dev), rtol=1e-5, atol=2e-3)
    # untouched particles come back to where they started
    still = (xb == x0).all(dim=1) & close
    assert bool(((xa[still].double() - x0[still].double()).abs() <= (1e-13 if xdt == "f64" else 1e-5) * (1 + x0[still].double().abs())).all())
    # adaptation on: rho_hist[t+1] = pcn_adapt(rho_hist[t], acc_t)
    xa, lla, lpa, lqa = init()
    n_acc, rho_hist, rho_out = eng.pcn_mutate_flow(xa, lla, lpa, lqa, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho,
                                                   n_steps, 5, 0.234, True, "f64", nu)
    r = rho
    for t in range(n_steps):
        assert rho_hist[t] == pytest.approx(r, rel=1e-12)
        r = pcn_adapt(rho_hist[t], n_acc[t] / n, 0.234, t)
    assert rho_out == pytest.approx(r, rel=1e-12)


@pytest.mark.gpu
def test_config3_flow_proposal_runs_on_the_mfma_path(eng):
    """BASELINE config 3 at reduced N: a trained float32 coupling-flow proposal, built-in targets, pCN mutation.
    The whole mutation loop must run in asmc_pcn_mutate_flow (flow log-density on the MFMA), and the evidence of
    the unnormalised Gaussian product must come out right."""
    from aspire_amd.flows import CouplingFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 32, 20000
    g = np.random.default_rng(0)
    flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
    flow.fit(1.3 * g.normal(size=(6000, d)), n_epochs=15)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(2))
    eng.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=8), store_sample_history=False)
    rep = eng.profile_report()
    eng.profile(False)
    # d = 32, single-Gaussian targets, resident flow: the whole step is ONE kernel (propose -> flow on the MFMA -> accept)
    assert rep["k_pcn_flow_fused"][0] == 8 * len(sp.history.beta)
    assert "k_pcn_propose" not in rep and "k_pcn_flow_propose" not in rep  # neither split path
    true = 0.5 * d * math.log(math.pi)
    assert sp.history.beta[-1] == 1.0
    assert abs(float(out.log_evidence) - true) < 5 * float(out.log_evidence_error) + 0.05, (float(out.log_evidence), true)
    assert 0.05 < np.mean(sp.history.mcmc_acceptance) < 0.95
    xs = out.x.double()
    assert float(xs.var(dim=0).mean()) == pytest.approx(0.5, rel=0.1)  # posterior N(0, I/2)


# ---- preconditioning transforms ---------------------------------------------------
@pytest.mark.gpu
def test_transform_kernels_match_reference_golden(eng, golden):
    """asmc_transform_forward / _inverse against the REAL reference's CompositeTransform outputs (golden
    ref_transforms.npz): values and log|det J| to 1e-12 (device libm vs numpy/scipy), and against the oracle."""
    import sys as _sys
    import os as _os
    _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
    from test_host_logic import _composite_from_golden

    g = golden["ref_transforms"]
    for name in g["names"]:
        T = _composite_from_golden(g, name, eng)
        x = eng.asarray(g[f"{name}_x"])
        z_fit = T.fit(x)
        np.testing.assert_allclose(eng.to_numpy(z_fit), g[f"{name}_z_fit"], rtol=1e-12, atol=1e-12)
        if int(g[f"{name}_affine"]):
            np.testing.assert_allclose(T._mean, g[f"{name}_mean"], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(T._std, g[f"{name}_std"], rtol=1e-12)
        z, lj = T.forward(x)
        np.testing.assert_allclose(eng.to_numpy(z), g[f"{name}_z"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(eng.to_numpy(lj), g[f"{name}_lj"], rtol=1e-12, atol=1e-11)
        x2, lj2 = T.inverse(eng.asarray(g[f"{name}_z2"]))
        np.testing.assert_allclose(eng.to_numpy(x2), g[f"{name}_x2"], rtol=1e-12, atol=1e-12)
        # inverse log-Jacobian: the reference's logit form, log(u) + log1p(-u) on the ROUNDED u = 1 / (1 + exp(-v)), loses up
        # to ~1e-10 to cancellation where u is next to 1; the kernel's -|v| - 2 log(1 + exp(-|v|)) does not
        np.testing.assert_allclose(eng.to_numpy(lj2), g[f"{name}_lj2"], rtol=1e-12, atol=2e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,xdt", [(1, 3, "f64"), (65, 5, "f64"), (10007, 32, "f64"), (4099, 32, "f32"), (300, 7, "f32")])
def test_transform_kernels_vs_oracle_ragged_and_fp32(eng, oracle, n, d, xdt):
    """Ragged sizes, odd row lengths (4/8-byte tile copies), fp32 storage, in-place operation; forward o inverse."""
    g = np.random.default_rng(n + d)
    kind = g.integers(0, 3, size=d).astype(np.int32)
    kind[kind == 1] = 1
    if (kind == 1).any() and (kind == 2).any():  # the reference uses ONE bounded transform per composite
        kind[kind == 2] = 1
    per = ((kind == 0) & (g.uniform(size=d) < 0.4)).astype(np.int32)
    lo = g.uniform(-3, 0, size=d)
    up = lo + g.uniform(0.5, 6, size=d)
    mean, std = g.normal(size=d), g.uniform(0.5, 2, size=d)
    x = np.where(per[None, :] == 1, lo + (up - lo) * g.uniform(-0.5, 1.5, size=(n, d)),
                 lo + (up - lo) * g.uniform(0.001, 0.999, size=(n, d)))
    dt = torch.float64 if xdt == "f64" else torch.float32
    xt = eng.asarray(x, dtype=dt)
    xr = xt.double().cpu().numpy()
    unit = float(-np.log((up - lo)[kind != 0]).sum()) if (kind != 0).any() else 0.0
    t = eng.make_transform(kind, per, lo, up, mean, std, 1e-6, unit, float(-np.log(np.abs(std)).sum()))
    z, lj = eng.transform_forward(xt, t)
    zr, ljr = oracle.transform(xr, kind, per, lo, up, mean, std, 1e-6)
    tol = dict(rtol=1e-12, atol=1e-12) if xdt == "f64" else dict(rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(z.double().cpu().numpy(), zr, **tol)
    np.testing.assert_allclose(lj.cpu().numpy(), ljr, rtol=1e-12, atol=1e-11)
    xb, ljb = eng.transform_inverse(z, t)
    xbr, ljbr = oracle.transform(z.double().cpu().numpy(), kind, per, lo, up, mean, std, 1e-6, inverse=True)
    np.testing.assert_allclose(xb.double().cpu().numpy(), xbr, **tol)
    np.testing.assert_allclose(ljb.cpu().numpy(), ljbr, rtol=1e-12, atol=1e-11)
    if xdt == "f64":  # forward then inverse is the identity away from the eps clamp, and the log-Jacobians cancel
        inside = per == 0
        np.testing.assert_allclose(xb.cpu().numpy()[:, inside], xr[:, inside], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose((lj + ljb).cpu().numpy(), 0.0, atol=1e-8)


@pytest.mark.gpu
def test_sampler_with_bounded_preconditioning_gpu(eng):
    """Aspire facade with prior bounds and preconditioning_kwargs(bounded_to_unbounded, affine): the mutation runs
    in the transformed space on the device; evidence of the bounded problem and containment of the particles."""
    from aspire_amd import Aspire
    from aspire_amd.flows import GaussianFlow
    from test_host_logic import _bounded_problem

    d, n = 4, 20000
    log_prior, log_like, true_logz = _bounded_problem(d)
    params = [f"x_{i}" for i in range(d)]
    asp = Aspire(log_likelihood=log_like, log_prior=log_prior, dims=d, parameters=params,
                 prior_bounds={p: (-4.0, 4.0) for p in params}, bounded_to_unbounded=False, xp=np,
                 flow=GaussianFlow(d, sigma=1.6, engine=eng, seed=3))
    out, hist = asp.sample_posterior(n, sampler="smc", return_history=True, rng=np.random.default_rng(4),
                                     preconditioning_kwargs=dict(bounded_to_unbounded=True, affine_transform=True),
                                     sampler_kwargs=dict(n_steps=6), store_sample_history=False, engine=eng)
    from aspire_amd.transforms import CompositeTransform
    assert isinstance(asp.sampler.preconditioning_transform, CompositeTransform)
    assert hist.beta[-1] == 1.0
    assert np.all(np.abs(np.asarray(out.x)) <= 4.0)
    assert abs(float(out.log_evidence) - true_logz) < 5 * float(out.log_evidence_error) + 0.05, (float(out.log_evidence), true_logz)
    assert np.all(np.abs(np.asarray(out.x).var(axis=0) - 1.0) < 0.1)


@pytest.mark.gpu
def test_device_bisection_stress_equals_host_driven_exact_search(eng):
    """asmc_find_beta (closed-form shifts + geometric progression + in-launch tail) against the host-driven search that
    evaluates every candidate with its exact maximum and direct exponentials: same beta* on heavy-tailed, nearly
    degenerate, tiny and large populations, late-stage beta0 and both tolerances."""
    from aspire_amd import smc_math

    g = np.random.default_rng(2024)
    mismatches, cases = [], 0
    for trial in range(40):
        n = int(g.choice([3, 17, 64, 300, 2048, 5000, 70001, 400000]))
        kind = trial % 5
        if kind == 0:    # Gaussian-like
            ll, lp, lq = -0.5 * g.chisquare(8, n), -0.5 * g.chisquare(8, n), -0.5 * g.chisquare(8, n) - 3.0
        elif kind == 1:  # heavy tails: a few huge log-likelihoods
            ll, lp, lq = g.standard_t(2, n) * 20.0, g.normal(size=n), g.normal(size=n)
        elif kind == 2:  # peaked likelihood: Delta of order 1e3-1e4
            ll, lp, lq = -np.abs(g.normal(size=n)) * 3e3, g.normal(size=n), g.normal(size=n) * 2
        elif kind == 3:  # nearly uniform weights (search jumps to 1)
            ll, lp, lq = 1e-3 * g.normal(size=n), np.zeros(n), np.zeros(n)
        else:            # some particles with zero likelihood
            ll, lp, lq = g.normal(size=n), g.normal(size=n), g.normal(size=n)
            ll[g.uniform(size=n) < 0.2] = -np.inf
        b0 = float(g.choice([0.0, 0.0, 0.013, 0.4, 0.93, 0.999]))
        tol = float(g.choice([1e-6, 1e-8]))
        target = float(g.choice([0.5, 0.3, 0.9]))
        tl, tp, tq = dev(eng, ll, lp, lq)
        b_dev, eff1, conv, rounds, n_nan, trip, trip_one = eng.find_beta(tl, tp, tq, b0, target, tol)
        assert conv and n_nan == 0

        def eff_fn(betas):
            return [smc_math.ess(s) / n for s in smc_math.global_stats(eng, Comm(), tl, tp, tq, b0, betas, n)]

        try:
            b_host, _, _ = smc_math.determine_beta(eff_fn, b0, adaptive=True, beta_step=float("nan"), min_beta_step=0.0,
                                                   max_beta_step=1.0, beta_tolerance=tol, adaptive_min_beta_step=False,
                                                   target=target, rate=1.0)
        except smc_math.BetaScheduleError:
            b_host = b0
        cases += 1
        if b_dev != b_host and not (b_host == b0 and b_dev <= b0 + tol):
            mismatches.append((trial, n, kind, b0, tol, target, b_dev, b_host))
    assert not mismatches, mismatches
    assert cases == 40


# ---- d = 64 / 128 pCN on the fp64 matrix cores (BASELINE config 5) ---------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d,C,dtype,noise", [(128, 2, torch.float64, "f64"), (64, 1, torch.float64, "f64"),
                                             (128, 2, torch.float64, "f32"), (128, 1, torch.float32, "f64")])
def test_pcn_mfma_d64_d128_vs_oracle(eng, oracle, d, C, dtype, noise):
    """asmc_pcn_mutate at d = 64 / 128 runs on v_mfma_f64_16x16x4_f64 (whitened state) — against the oracle's x-space
    restatement over 3 steps: same accept decisions up to razor-edge cases, positions to 1e-9 (fp64 storage) /
    2e-5 (fp32), carried log-probabilities equal to the densities at the stored positions."""
    n = 1999  # ragged last group of 16
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 41 + d, C=C)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    xd = torch.as_tensor(x).to(dtype).to(eng.device)
    xr = xd.double().cpu().numpy().copy()
    ll, lp, lq = (m.logpdf(xr) for m in om)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    rho, beta, seed = 0.15, 0.6, 4321
    eng.profile(True)
    n_acc, rho_hist, _ = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0],
                                        dm[1], dm[2], seed, 50, rho, 3, 10, 0.234, False, noise)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_pcn_mm_step"][0] == 3 and "k_pcn_step_generic" not in rep
    llr, lpr, lqr = ll.copy(), lp.copy(), lq.copy()
    acc_ref = [oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, om[0], om[1], om[2], seed, 50, 10 + t, noise)
               for t in range(3)]
    got = xd.double().cpu().numpy()
    tol = 1e-9 if (dtype == torch.float64 and noise == "f64") else 3e-5
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    edge = 3 if (dtype == torch.float64 and noise == "f64") else 40
    assert (~close).sum() <= edge, (~close).sum()
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= edge)
    assert 0.02 < np.mean(n_acc) / n < 0.98
    np.testing.assert_allclose(lld.cpu().numpy(), om[0].logpdf(got), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lqd.cpu().numpy(), om[2].logpdf(got), rtol=1e-10, atol=1e-9)


@pytest.mark.gpu
def test_importance_sampler_default_path_gpu(eng):
    """`sample_posterior()` with the reference's default sampler ("importance") on the device: evidence of the
    Gaussian product within its own error bar, weights consistent with the returned log-probabilities."""
    from aspire_amd import Aspire
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 8, 200000
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    asp = Aspire(log_likelihood=lik, log_prior=lik, dims=d, xp=np, flow=GaussianFlow(d, sigma=1.2, engine=eng, seed=9))
    out = asp.sample_posterior(n, engine=eng)  # sampler defaults to "importance" (aspire.py:383-395)
    true = 0.5 * d * math.log(math.pi)
    assert abs(float(out.log_evidence) - true) < 5 * float(out.log_evidence_error) + 1e-3
    lw = np.asarray(out.log_likelihood) + np.asarray(out.log_prior) - np.asarray(out.log_q)
    np.testing.assert_allclose(np.asarray(out.log_w), lw, rtol=1e-13, atol=1e-12)
    assert 0 < float(out.effective_sample_size) <= n


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gaussian", "coupling"])
def test_aspire_prior_bounds_default_flow_transform_gpu(eng, backend):
    """Aspire(prior_bounds=...) with the reference's defaults (bounded_to_unbounded=True): the flow lives behind a
    FlowTransform on the device; proposal draws respect the box, log q carries the Jacobian, SMC gets the evidence."""
    from aspire_amd import Aspire, Samples
    from test_host_logic import _bounded_problem

    d = 4
    log_prior, log_like, true_logz = _bounded_problem(d)
    params = [f"x_{i}" for i in range(d)]
    kw = dict(engine=eng, seed=5) if backend == "gaussian" else dict(n_layers=2, hidden_features=(32, 32), seed=3)
    asp = Aspire(log_likelihood=log_like, log_prior=log_prior, dims=d, parameters=params, device="cuda",
                 prior_bounds={p: (-4.0, 4.0) for p in params}, flow_backend=backend, xp=np, **kw)
    g = np.random.default_rng(0)
    asp.fit(Samples(np.clip(1.2 * g.normal(size=(4000, d)), -3.9, 3.9), parameters=params, xp=np),
            **({} if backend == "gaussian" else dict(n_epochs=8)))
    x, lq = asp.flow.sample_and_log_prob(5000)
    assert bool((x.abs() < 4.0).all()) and bool(torch.isfinite(lq).all())
    torch.testing.assert_close(asp.flow.log_prob(x).double(), lq.double(), rtol=1e-4, atol=2e-3)
    post = asp.sample_posterior(20000, sampler="smc", engine=eng, rng=np.random.default_rng(1), sampler_kwargs=dict(n_steps=6),
                                store_sample_history=False)
    assert np.all(np.abs(np.asarray(post.x)) <= 4.0)
    assert abs(float(post.log_evidence) - true_logz) < 5 * float(post.log_evidence_error) + 0.05, (float(post.log_evidence), true_logz)


# ---- sharded building blocks on one GPU: rank records, select kernel ----------------------
@pytest.mark.parametrize("n,seed,beta0,target", [(4096, 3, 0.0, 0.5), (300001, 21, 0.0, 0.5), (1 << 20, 22, 0.2, 0.7),
                                                 (50000, 23, 0.0, 0.01)])
@pytest.mark.parametrize("split", [1, 2, 3])
def test_find_beta_shard_rounds_match_single_rank_search(eng, n, seed, beta0, target, split):
    """`split` emulated ranks (separate contexts on the one device, records concatenated by hand in rank order): the
    merged search takes the single-rank kernel's decisions."""
    from aspire_amd.engine import HipEngine

    x, ll, lp, lq = synth(n, 4, seed)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    want = eng.find_beta(lld, lpd, lqd, beta0, target, 1e-6)
    engines = [eng] + [HipEngine(0, n_max=n, d_max=32) for _ in range(split - 1)]
    cuts = [r * n // split for r in range(split + 1)]
    parts = [tuple(t[cuts[r]:cuts[r + 1]].contiguous() for t in (lld, lpd, lqd)) for r in range(split)]
    recs = [e.empty(40) for e in engines]
    rounds = max(1, math.ceil(math.log2((1.0 - beta0) / 1e-6) / 4 - 1e-9))
    for rnd in range(rounds + 1):  # one more than needed: rounds after convergence are no-ops
        for e, p, rec in zip(engines, parts, recs):
            e.find_beta_shard_reduce(*p, beta0, rnd, rec)
        allrec = torch.cat(recs).contiguous()
        for e in engines:
            e.find_beta_shard_decide(allrec, split, n, beta0, target, 1e-6, rnd)
    outs = [e.find_beta_shard_result() for e in engines]
    for got in outs:
        assert got[0] == want[0] and got[2] and got[3] == want[3]  # beta*, converged, rounds
        assert got[1] == pytest.approx(want[1], rel=1e-11)
        assert (got[5] is None) == (want[5] is None)
        if got[5] is not None:
            # local-maximum shifts rescaled to the merged one: same triple up to rounding of the shift
            assert got[5][0] == pytest.approx(want[5][0], rel=1e-13, abs=1e-13)
            assert got[5][1] == pytest.approx(want[5][1], rel=1e-11) and got[5][2] == pytest.approx(want[5][2], rel=1e-11)
        assert got[6][1] == pytest.approx(want[6][1], rel=1e-11)
    assert all(o == outs[0] for o in outs)  # identical on every rank
    bad = eng.asarray(np.array([0.0, np.nan, 1.0, 2.0]))
    z = eng.asarray(np.zeros(4))
    rec = eng.empty(40)
    eng.find_beta_shard_reduce(bad, z, z, 0.0, 0, rec)
    eng.find_beta_shard_decide(rec, 1, 4, 0.0, 0.5, 1e-6, 0)
    assert eng.find_beta_shard_result()[4] == 1  # NaN census travels in the record


@pytest.mark.parametrize("n_total", [1, 63, 1000, 65536, 65537, 300001, 1 << 21])
def test_pcg64_select_matches_specification(eng, oracle, n_total):
    from aspire_amd.smc_math import pcg64_state

    for seed, lo, hi in ((0, 0.0, 1.0), (5, 0.0, 0.37), (7, 0.37, 0.9000001), (9, 0.75, 1.0), (11, 0.5, 0.5 + 1e-9)):
        st = pcg64_state(np.random.default_rng(seed))
        q = eng.pcg64_select(st, n_total, lo, hi).cpu().numpy()
        want = oracle.pcg64_select(st, n_total, lo, hi)
        assert np.array_equal(q, want)
        assert q.size == 0 or (q.min() >= 0.0 and q.max() < 1.0)
    # the slices of a partition of [0, 1) keep every draw exactly once
    st = pcg64_state(np.random.default_rng(3))
    cuts = [0.0, 0.124, 0.5, 0.500001, 1.0]
    assert sum(eng.pcg64_select(st, n_total, a, b).numel() for a, b in zip(cuts[:-1], cuts[1:])) == n_total


def test_device_resident_partials_match_host_variants(eng):
    n = 200003
    x, ll, lp, lq = synth(n, 4, 31)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    from aspire_amd import smc_math
    from aspire_amd.comm import Comm

    st = smc_math.global_stats(eng, Comm(), lld, lpd, lqd, 0.0, [0.2], n)[0]
    shift = float((st.m + np.log(st.S1)) - math.log(n))
    mp_ = st.m + shift
    host = eng.weights_m2_lse(lld, lpd, lqd, 0.0, 0.2, st.m, st.S1 / n, shift, mp_)
    rec = eng.empty(4)
    eng.weights_m2_lse_dev(lld, lpd, lqd, 0.0, 0.2, st.m, st.S1 / n, shift, mp_, rec)
    w = eng.normalized_weights(lld, lpd, lqd, 0.0, 0.2, shift, mp_ + math.log(st.S1))
    _, total = eng.cdf(w, "exact", 0.0)
    eng.cdf_total_dev(rec[2:])
    got = rec.cpu().numpy()
    assert got[0] == host[0] and got[1] == host[1] and got[2] == total


def test_pcg64_select_at_eight_rank_scale(eng, oracle):
    """BASELINE config 4: 8 ranks x 1M particles -> every rank walks 8M draws and keeps its eighth."""
    from aspire_amd.smc_math import pcg64_state

    n_total = 8_000_000
    st = pcg64_state(np.random.default_rng(17))
    g = np.random.default_rng(1)
    t = 1.0 + 1e-3 * g.normal(size=8)  # rank totals: equal shares +- 0.1 %
    edges = np.concatenate([[0.0], np.cumsum(t)])
    cuts = edges / edges[-1]
    cuts[0], cuts[-1] = 0.0, 1.0
    sizes = []
    for r in range(8):
        q = eng.pcg64_select(st, n_total, float(cuts[r]), float(cuts[r + 1]))
        sizes.append(q.numel())
        if r in (0, 5):
            assert np.array_equal(q.cpu().numpy(), oracle.pcg64_select(st, n_total, float(cuts[r]), float(cuts[r + 1])))
    assert sum(sizes) == n_total
    assert max(abs(s - n_total / 8) for s in sizes) < 0.01 * n_total / 8


# ---- t-preconditioned Crank-Nicolson (step_fn "tpcn": Student-t reference; own specification, oracle orc_tpcn_step) ----
@pytest.mark.parametrize("d,C,n_steps,noise,nu", [(32, 1, 1, "f64", 7.5), (32, 1, 4, "f64", 3.0), (4, 2, 1, "f64", 1.0),
                                                  (7, 1, 1, "f64", 12.0), (8, 1, 4, "f32", 5.0), (64, 2, 3, "f64", 6.0),
                                                  (128, 1, 3, "f32", 9.0)])
def test_tpcn_step_vs_oracle(eng, oracle, d, C, n_steps, noise, nu):
    """Every kernel family (register-resident x / whitened state, generic LDS, fp64 matrix cores) against the oracle's
    restatement of the tpCN step: same gamma scale variates, proposals to 1e-9, accept decisions up to razor edges."""
    n = 2999
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 91 + d, C=C)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    xd, lld, lpd, lqd = dev(eng, x, ll, lp, lq)
    rho, beta, seed, gid0, step0 = (0.4 if d <= 32 else 0.15), 0.45, 9001, 77, 3
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0], dm[1],
                                 dm[2], seed, gid0, rho, n_steps, step0, 0.234, False, noise, nu)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_gamma_draw"][0] == -(-n_steps // 8)  # one launch draws the variates of eight steps
    if d in (64, 128):
        assert rep["k_tpcn_mm_step"][0] == n_steps
    elif d in (4, 8, 16, 32):
        assert rep["k_tpcn_reg_y" if (n_steps >= 4 and C == 1) else "k_tpcn_reg"][0] == n_steps
    xr, llr, lpr, lqr = x.copy(), ll.copy(), lp.copy(), lq.copy()
    acc_ref = [oracle.tpcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, om[0], om[1], om[2], seed, gid0, step0 + t, noise)
               for t in range(n_steps)]
    got = xd.cpu().numpy()
    tol = 1e-9 if noise == "f64" else 3e-5
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    edge = 3 if noise == "f64" else 40
    assert (~close).sum() <= edge, (~close).sum()
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= edge)
    assert 0.02 < np.mean(n_acc) / n < 0.98
    np.testing.assert_allclose(lld.cpu().numpy(), om[0].logpdf(got), rtol=1e-10, atol=1e-9)
    # the Student-t scale really is in play: a Gaussian-reference step from the same state proposes elsewhere
    xg = eng.asarray(x)
    eng.pcn_mutate(xg, *dev(eng, ll, lp, lq), beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0], dm[1], dm[2],
                   seed, gid0, rho, n_steps, step0, 0.234, False, noise, 0.0)
    assert not torch.equal(xg, xd)


def test_tpcn_split_path_equals_fused(eng, oracle):
    n, d, nu = 2000, 8, 4.5
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 78)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    ll, lp, lq = (m.logpdf(x) for m in om)
    a = dev(eng, x, ll, lp, lq)
    b = dev(eng, x, ll, lp, lq)
    mud, Ld, Lid = dev(eng, mu, L, Linv)
    n_acc, _, _ = eng.pcn_mutate(*a, 0.6, mud, Ld, Lid, dm[0], dm[1], dm[2], 5, 0, 0.3, 1, 9, 0.234, False, "f64", nu)
    xp, q0, q1 = eng.pcn_propose(b[0], mud, Ld, Lid, 0.3, 5, 0, 9, nu=nu)
    lln, lpn, lqn = (eng.mixture_logpdf(xp, m) for m in dm)
    nacc = eng.pcn_accept(b[0], xp, b[1], b[2], b[3], lln, lpn, lqn, q0, q1, 0.6, 5, 0, 9)
    # the generic propose kernel accumulates |y|^2 in the opposite order to the register kernel: the scale sqrt(s)
    # differs in the last bit, so positions agree to rounding instead of bitwise (they are bitwise equal for pCN)
    assert abs(nacc - int(n_acc[0])) <= 1
    far = (torch.abs(a[0] - b[0]) > 1e-12 * (1 + torch.abs(b[0]))).any(dim=1)
    assert far.sum().item() <= 2
    y = (x - mu) @ Linv.T
    np.testing.assert_allclose(q0.cpu().numpy(), 2.0 * oracle.tpcn_corr((y * y).sum(1), d, nu), rtol=1e-12)


def test_tpcn_scale_variates_are_inverse_gamma(eng):
    """Proposal law through the split path (no accept step): from y = 0 with rho = 1 and the identity reference,
    x' = sqrt(s) xi, s = nu / (2 g), g ~ Gamma((d + nu)/2)  =>  |x'|^2 (d + nu) / (nu d) ~ F(d, d + nu)."""
    from scipy import stats

    n, d, nu = 200000, 4, 6.0
    eye = eng.asarray(np.eye(d))
    xp, q0, q1 = eng.pcn_propose(eng.asarray(np.zeros((n, d))), eng.asarray(np.zeros(d)), eye, eye, 1.0, 5, 0, 0, nu=nu)
    r2 = (xp.cpu().numpy() ** 2).sum(1)
    assert stats.kstest(r2 * (d + nu) / (nu * d), stats.f(d, d + nu).cdf).pvalue > 1e-3
    assert float(q0.abs().max()) == 0.0  # (d + nu) log(1 + 0 / nu)
    np.testing.assert_allclose(q1.cpu().numpy(), (d + nu) * np.log1p(r2 / nu), rtol=1e-12)


def test_tpcn_leaves_gaussian_target_invariant(eng):
    n, d, nu = 200000, 4, 4.0
    g = np.random.default_rng(1)
    x = g.normal(size=(n, d)) * np.sqrt(0.5)
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
    xd = eng.asarray(x)
    ll = eng.mixture_logpdf(xd, tgt)
    lp = ll.clone()
    lq = eng.mixture_logpdf(xd, q)
    L = np.diag(np.full(d, 0.9))
    n_acc, rho_hist, rho = eng.pcn_mutate(xd, ll, lp, lq, 1.0, eng.asarray(np.full(d, 0.2)), eng.asarray(L),
                                          eng.asarray(np.linalg.inv(L)), tgt, tgt, q, 11, 0, 0.5, 40, 0, 0.234, True, "f64", nu)
    xs = xd.cpu().numpy()
    assert np.all(np.abs(xs.mean(0)) < 0.01) and np.all(np.abs(xs.var(0) - 0.5) < 0.01)
    from scipy import stats

    assert stats.kstest(xs[:, 0] / np.sqrt(0.5), "norm").pvalue > 1e-4  # still Gaussian, not t
    np.testing.assert_allclose(ll.cpu().numpy(), -0.5 * (xs**2).sum(1), rtol=1e-12)


def test_tpcn_default_sampler_run_gpu(eng):
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 32, 200000
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, engine=eng, seed=3), xp=np,
                engine=eng, rng=np.random.default_rng(4))
    out = sp.sample(n, sampler_kwargs=dict(n_steps=16, noise="f32"), store_sample_history=False)
    assert sp.sampler_kwargs["step_fn"] == "tpcn" and len(sp.history.mcmc_nu) == len(sp.history.beta)
    assert abs(float(out.log_evidence) - 0.5 * d * math.log(math.pi)) < 5 * float(out.log_evidence_error) + 0.02


@pytest.mark.parametrize("d,step_fn", [(8, "pcn"), (32, "tpcn"), (6, "pcn")])
def test_callable_density_sampler_run_gpu(eng, d, step_fn):
    """Arbitrary torch callables as likelihood / prior: whitened-state split session where the dimension has kernels
    (d = 8, 32), propose / accept otherwise (d = 6); evidence against the closed form either way."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC

    n = 100000
    f = lambda smp: -0.5 * (smp.x * smp.x).sum(1)  # noqa: E731
    eng.profile(True)
    sp = HipSMC(log_likelihood=f, log_prior=f, dims=d, prior_flow=GaussianFlow(d, sigma=1.5, engine=eng, seed=3), xp=torch,
                engine=eng, rng=np.random.default_rng(4))
    out = sp.sample(n, sampler_kwargs=dict(n_steps=12, step_fn=step_fn), store_sample_history=False)
    rep = eng.profile_report()
    eng.profile(False)
    assert ("k_pcn_flow_accept" in rep) == (d in (8, 32)) and ("k_pcn_accept_flags" in rep) == (d == 6)
    assert abs(float(out.log_evidence) - 0.5 * d * math.log(math.pi)) < 5 * float(out.log_evidence_error) + 0.02
    assert 0.05 < np.mean(sp.history.mcmc_acceptance) < 0.99


@pytest.mark.parametrize("d,nu,dtype", [(20, 0.0, torch.float64), (31, 0.0, torch.float64), (17, 5.0, torch.float64),
                                        (24, 0.0, torch.float32), (6, 0.0, torch.float64), (12, 5.0, torch.float64),
                                        (48, 0.0, torch.float64), (64, 0.0, torch.float64), (64, 5.0, torch.float32),
                                        (100, 5.0, torch.float64), (100, 0.0, torch.float32), (128, 0.0, torch.float64),
                                        (128, 4.0, torch.float64)])
def test_pcn_propose_padded_dims_vs_oracle_engine(eng, oracle, d, nu, dtype):
    """Every d <= 128 proposes on a fast kernel: 16 < d < 32 on the 32-dimensional register kernel with identity-padded tables
    inside the kernel, d = 64 / 128 on the fp64 matrix cores (k_pcn_mm_propose), every other d on zero-padded copies of the
    rows through the kernel of the next width; proposals and correction terms agree with the test double's restatement (same
    noise, same scale variates) and the generic LDS kernel is never launched."""
    from oracle_engine import OracleEngine

    n = 3001
    x, mu, L, Linv, _ = _pcn_setup(eng, n, d, 300 + d)
    xd = torch.as_tensor(x).to(dtype).to(eng.device)
    eng.profile(True)
    xp, q0, q1 = eng.pcn_propose(xd, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), 0.35, 99, 1000, 7, nu=nu)
    rep = eng.profile_report()
    eng.profile(False)
    assert ("k_pcn_mm_propose" if d > 32 else "k_pcn_propose_reg") in rep and "k_pcn_propose" not in rep
    assert ("k_pad_rows" in rep) == (d not in (64, 128) and not 16 < d < 32)
    ref = OracleEngine()
    xr, r0, r1 = ref.pcn_propose(xd.double().cpu(), torch.as_tensor(mu), torch.as_tensor(L), torch.as_tensor(Linv), 0.35, 99,
                                 1000, 7, nu=nu)
    tol = 1e-11 if dtype == torch.float64 else 2e-6
    np.testing.assert_allclose(xp.double().cpu().numpy(), xr.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(q0.cpu().numpy(), r0.numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(q1.cpu().numpy(), r1.numpy(), rtol=1e-9 if dtype == torch.float64 else 1e-4, atol=1e-9)
    assert torch.equal(xd, torch.as_tensor(x).to(dtype).to(eng.device))  # the input is not touched


def test_shard_entry_points_reject_bad_arguments(eng):
    """Error convention of the sharded entry points: negative return code -> AsmcError with asmc_last_error()'s text."""
    from aspire_amd._lib import AsmcError
    from aspire_amd.smc_math import pcg64_state

    st = pcg64_state(np.random.default_rng(0))
    for lo, hi in ((0.5, 0.5), (0.6, 0.4), (-0.1, 0.5), (0.2, 1.5)):
        with pytest.raises(AsmcError, match="lo < hi"):
            eng.pcg64_select(st, 1000, lo, hi)
    z = eng.asarray(np.zeros(8))
    rec = eng.empty(40)
    with pytest.raises(AsmcError, match="beta0"):
        eng.find_beta_shard_reduce(z, z, z, 1.0, 0, rec)
    with pytest.raises(AsmcError, match="tolerance"):
        eng.find_beta_shard_decide(rec, 1, 8, 0.0, 0.5, 0.0, 0)
    with pytest.raises(AsmcError, match="nu must be"):
        x = eng.asarray(np.zeros((8, 4)))
        eye = eng.asarray(np.eye(4))
        eng.pcn_propose(x, eng.asarray(np.zeros(4)), eye, eye, 0.3, 1, 0, 0, nu=0.5)


@pytest.mark.parametrize("m,d", [(2048, 32), (1999, 7), (4096, 128), (64, 2), (1000, 5), (16384, 128), (700, 13)])
def test_student_fit_kernels_vs_numpy(eng, m, d):
    """asmc_student_estep / asmc_student_scale against their numpy restatement, then the whole device-driven EM against the
    all-numpy EM on heavy-tailed data."""
    from oracle_engine import OracleEngine
    from aspire_amd.student_t import fit_student_t, fit_student_t_device

    g = np.random.default_rng(m + d)
    A = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.linalg.cholesky(A @ A.T + 0.5 * np.eye(d))
    x = 0.3 + (g.normal(size=(m, d)) @ L.T) / np.sqrt(g.chisquare(6.0, size=m) / 6.0)[:, None]
    mu = x.mean(0) + 0.01
    Linv = np.linalg.inv(L)
    xd = eng.asarray(x)
    z, sz, sl, szx = eng.student_estep(xd, mu, np.tril(Linv), 7.5)
    zr, szr, slr, szxr = OracleEngine().student_estep(torch.as_tensor(x), mu, np.tril(Linv), 7.5)
    np.testing.assert_allclose(z.cpu().numpy(), zr.numpy(), rtol=1e-12)
    assert sz == pytest.approx(szr, rel=1e-12) and sl == pytest.approx(slr, rel=1e-11)
    np.testing.assert_allclose(szx, szxr, rtol=1e-11, atol=1e-11)
    r = eng.student_scale(xd, z, mu)
    np.testing.assert_allclose(r.cpu().numpy(), np.sqrt(zr.numpy())[:, None] * (x - mu), rtol=1e-13, atol=1e-14)
    if m > d + 10:
        a = fit_student_t(x, max_iter=8)
        b = fit_student_t_device(eng, xd, max_iter=8)
        np.testing.assert_allclose(b[0], a[0], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(b[1], a[1], rtol=1e-7, atol=1e-9)
        assert b[2] == pytest.approx(a[2], rel=1e-6)


@pytest.mark.parametrize("d,C,nu,noise,dtype", [(32, 2, 0.0, "f64", torch.float64), (8, 3, 0.0, "f64", torch.float64),
                                                (32, 2, 6.0, "f64", torch.float64), (16, 2, 0.0, "f32", torch.float32)])
def test_pcn_whitened_state_with_mixture_targets_vs_oracle(eng, oracle, d, C, nu, noise, dtype):
    """Several mixture components on the coordinate-major whitened state (PCN_Y_STEP_SG / _TSG: x' materialised in a second
    register array): 4 steps against the oracle's x-space restatement."""
    n = 3000
    x, mu, L, Linv, mixes = _pcn_setup(eng, n, d, 500 + d, C=C)
    om = [oracle.Mixture(*m) for m in mixes]
    dm = [eng.make_mixture(*m) for m in mixes]
    xd = torch.as_tensor(x).to(dtype).to(eng.device)
    xr = xd.double().cpu().numpy().copy()
    ll, lp, lq = (m.logpdf(xr) for m in om)
    lld, lpd, lqd = dev(eng, ll, lp, lq)
    rho, beta, seed = 0.35, 0.6, 1777
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), dm[0], dm[1], dm[2],
                                 seed, 50, rho, 4, 10, 0.234, False, noise, nu)
    rep = eng.profile_report()
    eng.profile(False)
    if not os.environ.get("ASMC_PCN_AOS"):  # (the row-major fallback runs mixtures on the x-state kernel)
        assert rep["k_tpcn_reg_y" if nu > 0 else "k_pcn_reg_y"][0] == 4 and "k_pcn_reg" not in rep
    llr, lpr, lqr = ll.copy(), lp.copy(), lq.copy()
    def step(t):
        if nu > 0:
            return oracle.tpcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, om[0], om[1], om[2], seed, 50, 10 + t, noise)
        return oracle.pcn_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, om[0], om[1], om[2], seed, 50, 10 + t, noise)

    acc_ref = [step(t) for t in range(4)]
    got = xd.double().cpu().numpy()
    exact = dtype == torch.float64 and noise == "f64"
    tol = 1e-9 if exact else 3e-5
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    edge = 3 if exact else 60
    assert (~close).sum() <= edge, (~close).sum()
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= edge)
    assert 0.02 < np.mean(n_acc) / n < 0.98
    np.testing.assert_allclose(lld.cpu().numpy(), om[0].logpdf(got), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lqd.cpu().numpy(), om[2].logpdf(got), rtol=1e-10, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("xdt,nu,noise,hidden,n", [("f64", 0.0, "f64", 64, 5000), ("f32", 0.0, "f64", 64, 4097), ("f64", 5.0, "f64", 64, 5000),
                                                   ("f64", 0.0, "f32", 64, 70001), ("f64", 0.0, "f64", 32, 3000), ("f64", 0.0, "f64", 128, 640),
                                                   ("f64", 0.0, "f64", 128, 100_000), ("f64", 0.0, "f64", 32, 100_000),
                                                   ("f64", 0.0, "f64", 64, 1)])
def test_pcn_flow_fused_step_vs_split_calls(eng, xdt, nu, noise, hidden, n):
    """The fused flow-proposal step (k_pcn_flow_fused: propose -> coupling flow on the MFMA -> built-in targets -> accept in
    one kernel; d = 32, single-Gaussian targets) against the same steps issued one ABI call at a time in x space, each of
    which is checked against the oracle above: same proposals (same Philox counters), same flow arithmetic (the
    stand-alone kernel's tile code), so positions agree to the rounding of the whitened state and accept decisions differ
    only at razor edges; carried log-probabilities equal the densities at the returned positions; ragged last tile."""
    from conftest import random_coupling_flow

    d, n_steps, beta, rho = 32, 5, 0.35, 0.4
    dt = torch.float64 if xdt == "f64" else torch.float32
    flow = random_coupling_flow(d, 4 if hidden < 128 else 1, hidden)  # every layer must be resident in LDS (W = 128: one fits)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(3)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g).to(dt)
    t_ll = eng.make_mixture([0.3], np.full((1, d), 0.25), np.ones((1, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(0.1 * np.arange(d) / d)
    A = np.eye(d) + 0.05 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    L, Linv = eng.asarray(A), eng.asarray(np.linalg.inv(A))

    def init():
        x = x0.clone()
        return x, eng.mixture_logpdf(x, t_ll), eng.mixture_logpdf(x, t_lp), eng.coupling_logprob(x, dev)

    xa, lla, lpa, lqa = init()
    eng.profile(True)
    n_acc, rho_hist, rho_out = eng.pcn_mutate_flow(xa, lla, lpa, lqa, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho,
                                                   n_steps, 5, 0.234, False, noise, nu)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_pcn_flow_fused"][0] == n_steps and "k_coupling_logprob" not in rep
    xb, llb, lpb, lqb = init()
    acc_b = []
    if noise == "f64":  # the split ABI calls draw fp64 noise
        for t in range(n_steps):
            xp, q0, q1 = eng.pcn_propose(xb, mu, L, Linv, rho, 77, 1000, 5 + t, nu=nu)
            lqn = eng.coupling_logprob(xp, dev)
            acc_b.append(eng.pcn_accept(xb, xp, llb, lpb, lqb, eng.mixture_logpdf(xp, t_ll), eng.mixture_logpdf(xp, t_lp), lqn,
                                        q0, q1, beta, 77, 1000, 5 + t))
    else:  # fast noise: the three-kernel device loop is the comparison (ASMC_FLOW_SPLIT is read per call)
        os.environ["ASMC_FLOW_SPLIT"] = "1"
        try:
            acc_b, _, _ = eng.pcn_mutate_flow(xb, llb, lpb, lqb, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho, n_steps, 5, 0.234,
                                              False, noise, nu)
        finally:
            del os.environ["ASMC_FLOW_SPLIT"]
        acc_b = acc_b.tolist()
    assert n == 1 or 0 < sum(acc_b) < n * n_steps
    tol = 1e-9 if xdt == "f64" else 2e-5
    close = ((xa.double() - xb.double()).abs() <= tol * (1 + xb.double().abs())).all(dim=1)
    edge = 5 if xdt == "f64" else 60
    assert int((~close).sum()) <= edge, int((~close).sum())
    assert np.all(np.abs(n_acc - np.array(acc_b)) <= edge)
    torch.testing.assert_close(lla, eng.mixture_logpdf(xa, t_ll), rtol=1e-9 if xdt == "f64" else 1e-4, atol=1e-9 if xdt == "f64" else 2e-3)
    torch.testing.assert_close(lpa, eng.mixture_logpdf(xa, t_lp), rtol=1e-9 if xdt == "f64" else 1e-4, atol=1e-9 if xdt == "f64" else 2e-3)
    torch.testing.assert_close(lqa, eng.coupling_logprob(xa, dev), rtol=1e-5, atol=2e-3)
    still = (xb == x0).all(dim=1) & close
    assert bool(((xa[still].double() - x0[still].double()).abs() <= (1e-13 if xdt == "f64" else 1e-5) * (1 + x0[still].double().abs())).all())


@pytest.mark.gpu
@pytest.mark.parametrize("nu", [0.0, 5.0, 2.5])
def test_pcn_flow_fused_step_vs_oracle(eng, oracle, nu):
    """The same fused step against the CPU oracle's restatement of the whole step (orc_pcn_flow_step: proposal, fp32
    coupling-flow log q, targets, accept): positions to 1e-9, accept decisions equal up to razor edges (the flow is fp32 on
    both sides with different summation orders: |delta log q| ~ 1e-5 moves a handful of decisions per 10^4).  nu > 0: the
    reference's DEFAULT step_fn, "tpcn" (smc/minipcn.py:46-49), composed with the flow proposal - k_pcn_flow_fused<..., TPCN>
    against orc_tpcn_flow_step_kind (round 4 compared that instantiation with the split HIP kernels only)."""
    from conftest import random_coupling_flow

    n, d, n_steps, beta, rho = 6000, 32, 3, 0.4, 0.35
    flow = random_coupling_flow(d, 4, 64)
    dev = flow.device_coupling(eng)
    ws, bs = flow.export_layers()
    g = np.random.default_rng(8)
    x0 = 0.9 * g.normal(size=(n, d))
    tgt_o = oracle.Mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    t_ll = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.linalg.cholesky(0.8 * (np.eye(d) + 0.2 * a @ a.T))
    Linv, mu = np.linalg.inv(L), 0.05 * g.normal(size=d)
    xr, llr = x0.copy(), tgt_o.logpdf(x0)
    lpr, lqr = llr.copy(), oracle.coupling_logprob(x0, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    xd = eng.asarray(x0)
    lld, lpd, lqd = eng.asarray(llr), eng.asarray(lpr), eng.coupling_logprob(xd, dev)
    np.testing.assert_allclose(lqd.cpu().numpy(), lqr, rtol=1e-5, atol=3e-4)  # two fp32 evaluations, different summation order
    from conftest import flow_log_prob_f64

    ref64 = flow_log_prob_f64(flow, x0)  # the bar itself: 1e-6 relative against the same flow in fp64
    assert np.max(np.abs(lqd.cpu().numpy() - ref64) / np.maximum(np.abs(ref64), 1.0)) <= 1e-6
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate_flow(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(np.tril(L)), eng.asarray(np.tril(Linv)),
                                      t_ll, t_ll, dev, 4242, 17, rho, n_steps, 9, 0.234, False, "f64", nu)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_pcn_flow_fused"][0] == n_steps and "k_coupling_logprob" not in rep, sorted(rep)
    acc_ref, margins = [], []
    for t in range(n_steps):
        with oracle.accept_margins(n) as m:
            if nu > 0:
                acc_ref.append(oracle.tpcn_flow_step(xr, llr, lpr, lqr, beta, mu, np.tril(L), np.tril(Linv), rho, nu, tgt_o, tgt_o, ws,
                                                     bs, flow.loc.numpy(), flow.scale.numpy(), 4242, 17, 9 + t, "f64", 0))
            else:
                acc_ref.append(oracle.pcn_flow_step(xr, llr, lpr, lqr, beta, mu, np.tril(L), np.tril(Linv), rho, tgt_o, tgt_o, ws, bs,
                                                    flow.loc.numpy(), flow.scale.numpy(), 4242, 17, 9 + t, "f64", 0))
        margins.append(m.copy())
    got = xd.cpu().numpy()
    close = np.all(np.abs(got - xr) <= 1e-9 * (1 + np.abs(xr)), axis=1)
    assert (~close).sum() <= 12, (~close).sum()
    # ... and they ARE razor edges: a row that ends elsewhere took a different decision at some step, and at its first such
    # step both sides held the same state - the restatement's accept margin log_a - log u there is within fp32 flow rounding
    razor = np.min(np.abs(np.array(margins)), axis=0)
    assert np.all(razor[~close] <= 1e-4), razor[~close]
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= 12) and 0.05 < np.mean(n_acc) / n < 0.95
    np.testing.assert_allclose(lld.cpu().numpy()[close], llr[close], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(lqd.cpu().numpy()[close], lqr[close], rtol=1e-5, atol=3e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("c_ll,c_lp,xdt", [(2, 1, "f64"), (3, 4, "f64"), (4, 2, "f32"), (1, 3, "f64")])
def test_pcn_flow_fused_step_with_mixture_targets_vs_oracle(eng, oracle, c_ll, c_lp, xdt):
    """Built-in MIXTURE targets (up to four components each: BASELINE config 5's two-component likelihood shape at d = 32) stay
    on the one-kernel flow-proposal step: the further components' quadratic forms come from the same matrix-core accumulators
    and are folded into a running log-sum-exp.  Against the oracle's restatement of the whole step (orc_pcn_flow_step with
    orc_diag_mixture_logpdf targets); round 3 sent any mixture target to the split propose / flow / accept kernels."""
    from conftest import random_coupling_flow

    n, d, n_steps, beta, rho = 5000, 32, 3, 0.45, 0.3
    dt = torch.float64 if xdt == "f64" else torch.float32
    flow = random_coupling_flow(d, 4, 64, seed=5)
    dev = flow.device_coupling(eng)
    ws, bs = flow.export_layers()
    g = np.random.default_rng(21)

    def mix(C, spread):
        logw = np.log(g.dirichlet(np.ones(C) * 3.0))
        return logw, spread * g.normal(size=(C, d)), 0.5 + g.random(size=(C, d))

    m_ll, m_lp = mix(c_ll, 0.8), mix(c_lp, 0.3)
    o_ll, o_lp = oracle.Mixture(*m_ll), oracle.Mixture(*m_lp)
    t_ll, t_lp = eng.make_mixture(*m_ll), eng.make_mixture(*m_lp)
    x0 = torch.as_tensor(0.9 * g.normal(size=(n, d))).to(dt)
    xr = x0.double().numpy().copy()
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.tril(np.linalg.cholesky(0.8 * (np.eye(d) + 0.2 * a @ a.T)))
    Linv, mu = np.tril(np.linalg.inv(L)), 0.05 * g.normal(size=d)
    llr, lpr = o_ll.logpdf(xr), o_lp.logpdf(xr)
    lqr = oracle.coupling_logprob(xr, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    xd = x0.to(eng.device).contiguous()
    lld, lpd, lqd = eng.asarray(llr), eng.asarray(lpr), eng.coupling_logprob(xd, dev)
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate_flow(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), t_ll, t_lp, dev,
                                      99, 40, rho, n_steps, 2, 0.234, False, "f64", 0.0)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_pcn_flow_fused"][0] == n_steps and "k_coupling_logprob" not in rep and "k_pcn_flow_propose" not in rep
    acc_ref, margins = [], []
    for t in range(n_steps):
        with oracle.accept_margins(n) as m:
            acc_ref.append(oracle.pcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, o_ll, o_lp, ws, bs, flow.loc.numpy(),
                                                flow.scale.numpy(), 99, 40, 2 + t, "f64", 0))
        margins.append(m.copy())
    got = xd.double().cpu().numpy()
    tol = 1e-9 if xdt == "f64" else 3e-5
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    edge = 12 if xdt == "f64" else 80
    assert (~close).sum() <= edge, (~close).sum()
    if xdt == "f64":
        razor = np.min(np.abs(np.array(margins)), axis=0)
        assert np.all(razor[~close] <= 1e-4), razor[~close]
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= edge) and 0.03 < np.mean(n_acc) / n < 0.97
    # carried densities = the mixtures at the returned positions
    np.testing.assert_allclose(lld.cpu().numpy(), o_ll.logpdf(got), rtol=1e-10 if xdt == "f64" else 1e-4, atol=1e-9 if xdt == "f64" else 2e-3)
    np.testing.assert_allclose(lpd.cpu().numpy(), o_lp.logpdf(got), rtol=1e-10 if xdt == "f64" else 1e-4, atol=1e-9 if xdt == "f64" else 2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("d,nu,xdt", [(8, 0.0, "f64"), (12, 0.0, "f64"), (16, 0.0, "f64"), (20, 0.0, "f64"), (30, 0.0, "f64"), (2, 0.0, "f64"),
                                      (16, 5.0, "f64"), (20, 0.0, "f32"), (10, 4.0, "f64")])
def test_pcn_flow_fused_step_below_32_dimensions_vs_oracle(eng, oracle, d, nu, xdt):
    """A flow-proposal mutation in fewer than 32 dimensions takes the ONE-kernel step too (round 4): the library pads rows and
    tables to 32 (identity beyond d, no noise there), the kernel places the rows of x' where the coupling tiles want their two
    halves.  Against the oracle's d-dimensional restatement of the whole step (pCN: orc_pcn_flow_step_kind; with
    nu, tpCN - the reference's default step_fn -: orc_tpcn_flow_step_kind); the trace shows one
    k_pcn_flow_fused per step and neither the propose / accept halves of d = 8 / 16 nor the x-state split path of the other d
    (round 3: 0.35 / 0.42 ms and 0.8 ms per step at 1M particles)."""
    from conftest import random_coupling_flow

    n, n_steps, beta, rho = 3000, 3, 0.45, 0.35
    dt = torch.float64 if xdt == "f64" else torch.float32
    flow = random_coupling_flow(d, 4, 64, seed=7)
    dev = flow.device_coupling(eng)
    ws, bs = flow.export_layers()
    g = np.random.default_rng(31 + d)
    tgt = ([0.1], 0.2 * g.normal(size=(1, d)), 0.7 + g.random(size=(1, d)))
    o_t, t_t = oracle.Mixture(*tgt), eng.make_mixture(*tgt)
    x0 = torch.as_tensor(0.9 * g.normal(size=(n, d))).to(dt)
    xr = x0.double().numpy().copy()
    a = g.normal(size=(d, d)) / np.sqrt(d)
    L = np.tril(np.linalg.cholesky(0.8 * (np.eye(d) + 0.2 * a @ a.T)))
    Linv, mu = np.tril(np.linalg.inv(L)), 0.05 * g.normal(size=d)
    llr = o_t.logpdf(xr)
    lpr = llr.copy()
    lqr = oracle.coupling_logprob(xr, ws, bs, flow.loc.numpy(), flow.scale.numpy())
    xd = x0.to(eng.device).contiguous()
    lld, lpd, lqd = eng.asarray(llr), eng.asarray(lpr), eng.coupling_logprob(xd, dev)
    eng.profile(True)
    n_acc, _, _ = eng.pcn_mutate_flow(xd, lld, lpd, lqd, beta, eng.asarray(mu), eng.asarray(L), eng.asarray(Linv), t_t, t_t, dev,
                                      123, 40, rho, n_steps, 2, 0.234, False, "f64", nu)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_pcn_flow_fused"][0] == n_steps, sorted(rep)
    assert not any(k.startswith(("k_coupling_logprob", "k_pcn_flow_propose", "k_pcn_flow_accept", "k_pcn_propose", "k_pcn_mm", "k_copy_flagged")) for k in rep), sorted(rep)
    acc_ref, margins = [], []
    for t in range(n_steps):
        with oracle.accept_margins(n) as m:
            if nu > 0:  # the reference's default step_fn with a flow proposal: orc_tpcn_flow_step_kind
                acc_ref.append(oracle.tpcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, nu, o_t, o_t, ws, bs, flow.loc.numpy(),
                                                     flow.scale.numpy(), 123, 40, 2 + t, "f64", 0))
            else:
                acc_ref.append(oracle.pcn_flow_step(xr, llr, lpr, lqr, beta, mu, L, Linv, rho, o_t, o_t, ws, bs, flow.loc.numpy(),
                                                    flow.scale.numpy(), 123, 40, 2 + t, "f64", 0))
        margins.append(m.copy())
    got = xd.double().cpu().numpy()
    # carried densities = the targets / the flow at the returned positions, whatever the reference measure
    np.testing.assert_allclose(lld.cpu().numpy(), o_t.logpdf(got), rtol=1e-10 if xdt == "f64" else 1e-4, atol=1e-9 if xdt == "f64" else 2e-3)
    torch.testing.assert_close(lqd, eng.coupling_logprob(xd, dev), rtol=1e-5, atol=2e-3)
    assert 0.03 < np.mean(n_acc) / n < 0.97
    tol = 1e-9 if xdt == "f64" else 3e-5
    close = np.all(np.abs(got - xr) <= tol * (1 + np.abs(xr)), axis=1)
    edge = 12 if xdt == "f64" else 80
    assert (~close).sum() <= edge, (~close).sum()
    if xdt == "f64":
        razor = np.min(np.abs(np.array(margins)), axis=0)
        assert np.all(razor[~close] <= 1e-4), razor[~close]
    assert np.all(np.abs(np.array(n_acc) - np.array(acc_ref)) <= edge)


# ---- split-fp16 MFMA flow arithmetic: accuracy of the operands, not of a reduced-precision flow --------------------
@pytest.mark.parametrize("hidden,scale_x", [(64, 1.0), (32, 3.0), (128, 0.2)])
def test_split_fp16_flow_is_as_accurate_as_the_fp32_mfma_chain(eng, hidden, scale_x, monkeypatch):
    """The flow's fp32 layers run as three fp16 MFMA products per K = 16 on (hi, lo) operand pairs.  Against the same
    flow evaluated in fp64 (torch, CPU) its error must be that of fp32 arithmetic: the fp32 MFMA chain's own rms error on
    in-distribution rows, within 4x of it on far-out rows, and inside the north-star's 1e-6 relative bar on
    log q wherever the fp32 chain is (tools/flow_accuracy.py prints the table, torch's fp32 evaluation included)."""
    from conftest import random_coupling_flow

    d, n = 32, 1 << 16
    n_layers = 4 if hidden < 128 else 1
    flow = random_coupling_flow(d, n_layers, hidden, seed=11)
    f64 = random_coupling_flow(d, n_layers, hidden, seed=11, dtype=torch.float64)
    f64.layers.load_state_dict(flow.layers.state_dict())  # the fp32 parameters, widened
    f64.loc, f64.scale = flow.loc.double(), flow.scale.double()
    g = np.random.default_rng(2)
    x = scale_x * g.normal(size=(n, d))
    x[:64] *= 4.0  # a few far-out rows: large activations
    with torch.no_grad():
        ref = f64.log_prob(torch.as_tensor(x)).numpy()
    dev = flow.device_coupling(eng)
    xd = eng.asarray(x)
    monkeypatch.setenv("ASMC_FLOW_MATH", "f32")
    lq32 = eng.coupling_logprob(xd, dev).cpu().numpy()
    monkeypatch.delenv("ASMC_FLOW_MATH")
    lqhs = eng.coupling_logprob(xd, dev).cpu().numpy()
    assert not np.array_equal(lq32, lqhs)  # two different instruction streams
    e32, ehs = np.abs(lq32 - ref), np.abs(lqhs - ref)
    assert np.all(np.isfinite(lqhs))
    rms = lambda e: float(np.sqrt(np.mean(e**2)))  # noqa: E731
    # in-distribution rows: the same error as the fp32 chain (both are dominated by fp32 accumulation rounding)
    assert rms(ehs[64:]) <= 1.25 * rms(e32[64:]) + 1e-7, (rms(ehs[64:]), rms(e32[64:]))
    # rows four times outside: the (hi, lo) pairs carry 2^-24 relative operand error where fp32 operands carry none
    assert rms(ehs[:64]) <= 4.0 * rms(e32[:64]) + 1e-5, (rms(ehs[:64]), rms(e32[:64]))
    rel32, relhs = np.max(e32 / np.maximum(np.abs(ref), 1.0)), np.max(ehs / np.maximum(np.abs(ref), 1.0))
    assert relhs < max(1e-6, 2.0 * rel32), (relhs, rel32)


# ---- bounded priors: log q(x') from the preconditioned coordinate, without the probit / erfinv round trip --------------
@pytest.mark.parametrize("d,dtype,unbounded,kind,affine", [
    (32, torch.float64, (), "probit", True), (8, torch.float64, (1, 5), "probit", True), (16, torch.float32, (), "probit", True),
    (32, torch.float64, (), "logit", False), (8, torch.float64, (0, 3), "logit", True), (16, torch.float32, (), "logit", False)])
def test_flow_log_prob_from_preconditioned_equals_the_round_trip(eng, d, dtype, unbounded, kind, affine):
    """GaussianFlow behind a FlowTransform and a CompositeTransform preconditioning built from the same prior bounds
    (what Aspire(prior_bounds=...) does): log q(T^-1(z)) evaluated as one premapped Gaussian pass over z equals
    flow.log_prob(T.inverse(z)) - including rows pushed beyond the eps clip of the unit interval."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.transforms import CompositeTransform, FlowTransform

    g = np.random.default_rng(12)
    names = [f"p{i}" for i in range(d)]
    bounds = {p: ([-np.inf, np.inf] if i in unbounded else [float(-3 - i % 3), float(4 + i % 5)]) for i, p in enumerate(names)}
    centre = np.array([0.0 if i in unbounded else 0.5 * (bounds[p][0] + bounds[p][1]) for i, p in enumerate(names)])
    Tf = FlowTransform(names, prior_bounds=bounds, bounded_transform=kind, engine=eng)
    flow = GaussianFlow(d, engine=eng, data_transform=Tf, dtype=dtype)
    flow.fit(centre + 1.2 * g.normal(size=(4000, d)).clip(-2.4, 2.4))
    T = CompositeTransform(names, prior_bounds=bounds, bounded_to_unbounded=True, bounded_transform=kind, affine_transform=affine,
                           engine=eng)
    T.fit(centre + 0.9 * g.normal(size=(3000, d)).clip(-3.0, 3.0))
    f = flow.log_prob_from_preconditioned(T)
    assert f is not None
    n = 20000
    z = eng.asarray((1.4 if affine else 3.0) * g.normal(size=(n, d)), dtype=dtype)
    z[:50] *= 4.0  # far rows: y beyond the clip of the forward transform
    x, logj = T.inverse(z)
    ref = flow.log_prob(eng.asarray(x, dtype=dtype)).cpu().numpy()
    got = f(z, eng.asarray(logj)).cpu().numpy()
    assert np.all(np.isfinite(got))
    tol = 1e-8 if dtype == torch.float64 else 2e-3  # fp32 rows: x' itself is rounded to fp32 on the round trip
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * 10)
    # not applicable: another eps, or another bounded stage
    assert flow.log_prob_from_preconditioned(CompositeTransform(names, prior_bounds=bounds, bounded_transform=kind, eps=1e-5,
                                                                engine=eng)) is None
    other = CompositeTransform(names, prior_bounds=bounds, bounded_transform="logit" if kind == "probit" else "probit", engine=eng)
    other.fit(centre + 0.5 * g.normal(size=(500, d)).clip(-2, 2))
    assert flow.log_prob_from_preconditioned(other) is None


@pytest.mark.parametrize("d,dtype,kind,affine,noise,nu", [
    (32, torch.float64, "logit", False, "f64", 0.0), (32, torch.float64, "probit", True, "f64", 5.0),
    (8, torch.float32, "logit", True, "f32", 0.0), (16, torch.float64, "probit", False, "f32", 0.0)])
def test_ysplit_propose_with_the_transform_inside_equals_the_separate_passes(eng, d, dtype, kind, affine, noise, nu):
    """asmc_pcn_ysplit_propose_tr (inverse transform, log|J| and - through the premap - log q(x') applied to the
    register-resident proposal) against asmc_pcn_ysplit_propose + asmc_transform_inverse + asmc_mixture_logpdf_premap on
    the same session: the same x', log|J| (other summation order) and log q."""
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.transforms import CompositeTransform, FlowTransform

    g = np.random.default_rng(21)
    n = 5000
    names = [f"p{i}" for i in range(d)]
    bounds = {p: [float(-2 - i % 3), float(3 + i % 4)] for i, p in enumerate(names)}
    centre = np.array([0.5 * (bounds[p][0] + bounds[p][1]) for p in names])
    Tf = FlowTransform(names, prior_bounds=bounds, bounded_transform=kind, engine=eng)
    flow = GaussianFlow(d, engine=eng, data_transform=Tf, dtype=dtype)
    flow.fit(centre + 1.1 * g.normal(size=(3000, d)).clip(-2.0, 2.0))
    T = CompositeTransform(names, prior_bounds=bounds, bounded_to_unbounded=True, bounded_transform=kind, affine_transform=affine,
                           engine=eng)
    x0 = eng.asarray(centre + 0.8 * g.normal(size=(n, d)).clip(-2.5, 2.5), dtype=dtype)
    z = eng.asarray(T.fit(x0), dtype=dtype)
    logq = flow.log_prob_from_preconditioned(T)
    assert logq is not None
    mu = eng.asarray(0.05 * g.normal(size=d))
    A = np.eye(d) * (1.0 if affine else 2.0) + 0.05 * np.tril(g.normal(size=(d, d)), -1)
    L, Linv = eng.asarray(A), eng.asarray(np.linalg.inv(A))
    sess = eng.pcn_ysplit_begin(z, 0.4, mu, L, Linv, 77, 1000, 0.5, 0.234, True, nu, noise)
    assert sess is not None
    t_dev = T._tables()[1]
    zp = eng.pcn_ysplit_propose(sess, 9)
    x_ref, lj_ref = T.inverse(zp)
    lq_ref = logq(zp, eng.asarray(lj_ref))
    x_f, lj_f, lq_f = eng.pcn_ysplit_propose_tr(sess, 9, t_dev, logq.fused_args)
    assert torch.equal(x_f, eng.asarray(x_ref, dtype=dtype))
    np.testing.assert_allclose(lj_f.cpu().numpy(), eng.asarray(lj_ref).cpu().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lq_f.cpu().numpy(), lq_ref.cpu().numpy(), rtol=1e-12, atol=1e-11)
    x_g, lj_g, none = eng.pcn_ysplit_propose_tr(sess, 9, t_dev, None)  # without the density
    assert none is None and torch.equal(x_g, x_f) and torch.equal(lj_g, lj_f)


def test_fused_flow_step_reports_non_finite_flow_densities(eng, monkeypatch):
    """A flow whose activations leave the fp16 operand range (|.| >= 65504) turns into NaN densities in the split-fp16
    layers: those proposals are rejected AND counted (asmc_pcn_flow_nonfinite), so the host can say so; the fp32 MFMA
    chain evaluates the same flow without loss."""
    from conftest import random_coupling_flow

    d, n = 32, 20000
    flow = random_coupling_flow(d, 4, 64)
    with torch.no_grad():
        lin = [m for m in flow.layers[0].net if isinstance(m, torch.nn.Linear)]
        lin[0].weight.mul_(1e5)  # weights stay inside the fp16 range, hidden activations of the first layer reach ~1e5
        lin[1].weight.mul_(1e-5)  # ... brought back to O(1) by the next layer: finite in fp32 arithmetic
    flow._version += 1
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(5)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))

    def run():
        x = x0.clone()
        ll = eng.mixture_logpdf(x, tgt)
        n_acc, _, _ = eng.pcn_mutate_flow(x, ll, ll.clone(), eng.coupling_logprob(x, dev), 0.5, mu, eye, eye, tgt, tgt, dev, 3, 0, 0.3,
                                          2, 0, 0.234, False, "f64")
        return int(n_acc.sum()), eng.pcn_flow_nonfinite()

    acc_hs, bad_hs = run()
    assert bad_hs > 0.3 * 2 * n and acc_hs < 0.7 * 2 * n  # many proposals overflow the operand pairs: rejected and counted
    lq = eng.coupling_logprob(x0, dev).cpu().numpy()  # the stand-alone kernel reports them as NaN, never as a wrong number
    assert np.isnan(lq).sum() > 0.3 * n
    monkeypatch.setenv("ASMC_FLOW_MATH", "f32")
    acc_32, bad_32 = run()
    assert bad_32 == 0 and acc_32 > 0
    assert np.all(np.isfinite(eng.coupling_logprob(x0, dev).cpu().numpy()))
    # weights beyond the range are refused when the flow is packed for the split-fp16 kernels
    monkeypatch.delenv("ASMC_FLOW_MATH")
    with torch.no_grad():
        lin[0].weight.mul_(10.0)
    flow._version += 1
    with pytest.raises(Exception):
        flow.device_coupling(eng)


@pytest.mark.parametrize("n,d,dt", [(100_003, 32, torch.float64), (65_536, 128, torch.float64), (40_000, 32, torch.float32),
                                    (5_001, 12, torch.float64), (70_000, 64, torch.float64)])
def test_mean_gram_equals_the_two_calls(eng, n, d, dt):
    """asmc_mean_gram (column sums -> centre on the device -> centred Gram matrix, one synchronisation) against asmc_colsum,
    the host's division and asmc_centered_gram: the same bits, for the fp64-MFMA shapes and for the ones that fall back."""
    g = torch.Generator(eng.device).manual_seed(n)
    x = (0.3 + torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)).to(dt)
    n_mean = n + 7  # the divisor is the caller's (the global population of a sharded run)
    s0 = eng.colsum(x)
    g0 = eng.centered_gram(x, s0 / n_mean)
    s1, g1 = eng.mean_gram(x, n_mean)
    assert np.array_equal(s0, s1) and np.array_equal(g0, g1)


def test_mutation_calls_count_the_nans_of_the_carried_log_q(eng):
    """The reference checks log q for NaN after every mutation (smc/minipcn.py).  asmc_pcn_mutate / asmc_pcn_mutate_flow count
    them on the device and hand the count back with their own results (asmc_pcn_lq_nan): equal to asmc_count_nonfinite of the
    array afterwards, for a clean batch and for one with poisoned rows.  (A NaN log q alone does not survive a step: the
    tempered log-probability of such a state counts as -inf, as in smc/base.py, and the first finite proposal replaces it;
    rows whose POSITION is NaN propose NaN and keep it.)"""
    from conftest import random_coupling_flow

    d, n = 32, 30000
    flow = random_coupling_flow(d, 4, 64)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(11)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
    q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
    mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
    for poisoned in (0, 37):
        for kind in ("flow", "mixture"):
            x = x0.clone()
            ll = eng.mixture_logpdf(x, tgt)
            lq = eng.coupling_logprob(x, dev) if kind == "flow" else eng.mixture_logpdf(x, q)
            lq[:poisoned] = float("nan")
            x[:poisoned] = float("nan")
            if kind == "flow":
                eng.pcn_mutate_flow(x, ll, ll.clone(), lq, 0.5, mu, eye, eye, tgt, tgt, dev, 3, 0, 0.3, 3, 0, 0.234, True, "f64")
            else:
                eng.pcn_mutate(x, ll, ll.clone(), lq, 0.5, mu, eye, eye, tgt, tgt, q, 3, 0, 0.3, 3, 0, 0.234, True, "f64")
            assert eng.pcn_lq_nan() == eng.count_nonfinite(lq)[0]
            assert (eng.pcn_lq_nan() > 0) == (poisoned > 0)


# ---- proposal draw from the coupling flow on the engine (asmc_coupling_sample) -----------------------------------------
@pytest.mark.parametrize("d,hidden,n_layers,dtype", [(32, 64, 4, torch.float64), (32, 32, 3, torch.float32), (16, 64, 4, torch.float64),
                                                      (6, 32, 2, torch.float64), (64, 32, 2, torch.float64)])
def test_coupling_sample_is_a_draw_from_the_flow(eng, d, hidden, n_layers, dtype):
    """The sampling kernel against the flow it samples: the reported log q equals the density kernel's (and torch's) at the
    emitted x, the latent recovered by the torch modules is standard normal, and the draw is keyed by the global particle
    index (a shard of the same draw is the same rows)."""
    from scipy import stats

    from conftest import random_coupling_flow

    flow = random_coupling_flow(d, n_layers, hidden, seed=7)
    dev = flow.device_coupling(eng)
    n = 40000
    x, lq = eng.coupling_sample(n, dtype, dev, 99, 1000, 3)
    assert x.dtype == dtype and tuple(x.shape) == (n, d)
    lq_k = eng.coupling_logprob(x, dev)
    np.testing.assert_allclose(lq.cpu().numpy(), lq_k.cpu().numpy(), rtol=2e-5, atol=2e-4)
    with torch.no_grad():
        z, _ = flow.forward(x.float().cpu())
        lq_t = flow.log_prob(x.float().cpu()).double().numpy()
    np.testing.assert_allclose(lq.cpu().numpy(), lq_t, rtol=1e-4, atol=2e-3)
    z = z.double().numpy()
    assert np.all(np.abs(z.mean(0)) < 0.03) and np.all(np.abs(z.var(0) - 1.0) < 0.05)
    assert stats.kstest(z[:, 0], "norm").pvalue > 1e-4 and stats.kstest(z[:, d - 1], "norm").pvalue > 1e-4
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03
    xs, lqs = eng.coupling_sample(100, dtype, dev, 99, 1000 + 777, 3)  # rows 777.. of the same draw
    assert torch.equal(xs, x[777:877]) and torch.equal(lqs, lq[777:877])
    x2, _ = eng.coupling_sample(100, dtype, dev, 99, 1000, 4)  # another draw id: other rows
    assert not torch.equal(x2, x[:100])


def test_coupling_flow_samples_on_the_engine_inside_the_sampler(eng):
    from aspire_amd.flows import CouplingFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 32, 100000
    flow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=12)
    flow.fit(1.3 * np.random.default_rng(3).normal(size=(4000, d)), n_epochs=4)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=np, engine=eng, rng=np.random.default_rng(2))
    eng.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=8, step_fn="pcn"), store_sample_history=False)
    rep = eng.profile_report()
    eng.profile(False)
    assert rep["k_coupling_sample"][0] >= 1
    assert abs(float(out.log_evidence) - 0.5 * d * math.log(math.pi)) < max(5 * float(out.log_evidence_error), 0.05)


@pytest.mark.parametrize("d", [4, 20, 32, 64, 100, 128])
def test_reference_factor_on_the_device_vs_numpy(eng, d):
    """asmc_reference_factor (mean, covariance, Cholesky factor and its inverse of the mutation's reference Gaussian in one
    block on the stream) against numpy on the same moments: from the host's (sums, Gram) for every d <= 128, from the pending
    asmc_mean_gram_enqueue for the matrix-core shapes; the jitter ladder on a singular covariance; -1 for a NaN."""
    n = 5000 + d
    g = torch.Generator(eng.device).manual_seed(d)
    A = torch.randn((d, d), device=eng.device, dtype=torch.float64, generator=g) / math.sqrt(d) + 0.7 * torch.eye(d, device=eng.device, dtype=torch.float64)
    x = (0.3 + torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g) @ A).contiguous()
    s, gram = eng.mean_gram(x, n)
    cov = gram / (n - 1)
    cov = 0.5 * (cov + cov.T)
    Lr = np.linalg.cholesky(cov)

    def check(mu, L, Linv):
        torch.cuda.synchronize()
        assert eng.reference_factor_status() == 0
        np.testing.assert_allclose(mu.cpu().numpy(), s / n, rtol=1e-15)
        Lh, Lih = L.cpu().numpy(), Linv.cpu().numpy()
        assert np.all(np.triu(Lh, 1) == 0) and np.all(np.triu(Lih, 1) == 0)
        np.testing.assert_allclose(Lh, Lr, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(Lih @ Lh, np.eye(d), atol=1e-11)

    check(*eng.reference_factor(d, n, n, moments=(s, gram)))
    if d in (32, 64, 128):
        assert eng.mean_gram_enqueue(x, n)
        eng.count_nonfinite(x[:, 0].contiguous())  # other launches use the same scratch: the moments have their own copy
        check(*eng.reference_factor(d, n, n))
        with pytest.raises(Exception, match="pending"):
            eng.mean_gram_fetch(d)  # consumed by the factorisation
        # the sharded form: sums and Gram matrix in the caller's device buffers (all-reduced there by the caller), same bits
        s_d = eng.colsum_dev(x)
        g_d = eng.centered_gram_dev(x, s_d, n)
        assert np.array_equal(s_d.cpu().numpy(), s) and np.array_equal(g_d.cpu().numpy(), gram)
        mu2, L2, Li2 = eng.reference_factor(d, n, n, moments_dev=(s_d, g_d))
        mu1, L1, Li1 = eng.reference_factor(d, n, n, moments=(s, gram))
        torch.cuda.synchronize()
        assert torch.equal(mu1, mu2) and torch.equal(L1, L2) and torch.equal(Li1, Li2)
    # a rank-deficient covariance needs the jitter; a NaN cannot be factored
    v = np.ones((d, 1))
    mu, L, Linv = eng.reference_factor(d, n, n, moments=(np.zeros(d), (n - 1) * (v @ v.T)))
    torch.cuda.synchronize()
    if d > 1:
        assert eng.reference_factor_status() >= 1
        Lh = L.cpu().numpy()
        np.testing.assert_allclose(Lh @ Lh.T, v @ v.T, atol=1e-6)
    bad = gram.copy()
    bad[0, 0] = np.nan
    eng.reference_factor(d, n, n, moments=(s, bad))
    torch.cuda.synchronize()
    assert eng.reference_factor_status() == -1


@pytest.mark.parametrize("d,nu", [(64, 0.0), (48, 0.0), (64, 6.0), (34, 0.0)])
def test_pcn_mutate_flow_above_32_dimensions_equals_split_calls(eng, d, nu, monkeypatch):
    """The FALLBACK of a neural proposal density above 32 dimensions (smc/base.py:507-519 is called for any `dims`): when the
    one-kernel step of round 5 (k_pcn_flow16, tests/test_gpu_flow16.py) cannot take a mutation - its tables exceed the LDS, or
    ASMC_FLOW16_OFF=1 as here - asmc_pcn_mutate_flow composes each step from the matrix-core propose kernel, the flow kernel,
    the targets and the accept / copy kernels on the x state (round 4's only path).  Those are the kernels behind the
    one-call-at-a-time ABI (propose / coupling_logprob / mixture_logpdf / accept), each checked against the oracle elsewhere: the
    device-side loop must return their bits - positions, carried log-probabilities, accept counts."""
    monkeypatch.setenv("ASMC_FLOW16_OFF", "1")
    from conftest import random_coupling_flow

    n, n_steps, beta, rho = 3000, 4, 0.35, 0.3
    flow = random_coupling_flow(d, 2, 64)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(5)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    t_ll = eng.make_mixture([0.0, -0.3], np.stack([np.full(d, 0.4), np.full(d, -0.4)]), np.ones((2, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(0.1 * np.arange(d) / d)
    A = np.eye(d) + 0.03 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    L, Linv = eng.asarray(A), eng.asarray(np.linalg.inv(A))

    def init():
        x = x0.clone()
        return x, eng.mixture_logpdf(x, t_ll), eng.mixture_logpdf(x, t_lp), eng.coupling_logprob(x, dev)

    xa, lla, lpa, lqa = init()
    eng.profile(True)
    n_acc, rho_hist, rho_out = eng.pcn_mutate_flow(xa, lla, lpa, lqa, beta, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, rho, n_steps, 5,
                                                   0.234, False, "f64", nu)
    rep = eng.profile_report()
    eng.profile(False)
    assert "k_pcn_flow_fused" not in rep and "k_pcn_flow16" not in rep and "k_tpcn_flow16" not in rep, sorted(rep)
    assert any(k.startswith(("k_coupling_logprob", "k_flow16_logprob")) for k in rep), sorted(rep)
    xb, llb, lpb, lqb = init()
    acc_b = []
    for t in range(n_steps):
        xp, q0, q1 = eng.pcn_propose(xb, mu, L, Linv, rho, 77, 1000, 5 + t, nu=nu)
        lqn = eng.coupling_logprob(xp, dev)
        acc_b.append(eng.pcn_accept(xb, xp, llb, lpb, lqb, eng.mixture_logpdf(xp, t_ll), eng.mixture_logpdf(xp, t_lp), lqn, q0, q1,
                                    beta, 77, 1000, 5 + t))
    assert 0 < sum(acc_b) < n * n_steps and n_acc.tolist() == acc_b and rho_out == rho
    for a, b in ((xa, xb), (lla, llb), (lpa, lpb), (lqa, lqb)):
        assert torch.equal(a, b)
    torch.testing.assert_close(lla, eng.mixture_logpdf(xa, t_ll), rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(lqa, eng.coupling_logprob(xa, dev), rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("hidden,n", [(64, 640), (64, 100_000), (128, 6400), (128, 100_000), (32, 100_000)])
def test_fused_flow_step_is_repeatable_and_carries_the_densities_of_its_positions(eng, hidden, n):
    """Twelve calls of the fused flow-proposal step on fresh copies of one batch, other kernels in between (tools/stress_fused.py
    in small): every call returns the bits of the first, and the carried log q / log-likelihood are the densities at the returned
    positions.  (The split-fp16 instantiation for hidden width 128 failed exactly this - with two interleaved flow tiles it spilled
    300 registers around the hand-scheduled flow code and returned different log q from run to run; that width now takes its
    tiles one at a time.)"""
    from conftest import random_coupling_flow

    d, n_steps = 32, 5
    flow = random_coupling_flow(d, 4 if hidden < 128 else 1, hidden)
    dev = flow.device_coupling(eng)
    g = torch.Generator(eng.device).manual_seed(3)
    x0 = torch.randn((n, d), device=eng.device, dtype=torch.float64, generator=g)
    t_ll = eng.make_mixture([0.3], np.full((1, d), 0.25), np.ones((1, d)) * 1.5)
    t_lp = eng.make_mixture([-0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.ones((1, d)))
    mu = eng.asarray(0.1 * np.arange(d) / d)
    A = np.eye(d) + 0.05 * np.tril(np.random.default_rng(2).normal(size=(d, d)), -1)
    L, Linv = eng.asarray(A), eng.asarray(np.linalg.inv(A))
    big = torch.randn((20000, 128), device=eng.device, dtype=torch.float64, generator=g)
    ll0, lp0, lq0 = eng.mixture_logpdf(x0, t_ll), eng.mixture_logpdf(x0, t_lp), eng.coupling_logprob(x0, dev)
    first = None
    for it in range(12):
        x, ll, lp, lq = x0.clone(), ll0.clone(), lp0.clone(), lq0.clone()
        if it % 3 == 1:
            eng.reference_factor(128, 20000, 20000, moments=eng.mean_gram(big, 20000))
        eng.profile(it == 0)
        n_acc, _, _ = eng.pcn_mutate_flow(x, ll, lp, lq, 0.35, mu, L, Linv, t_ll, t_lp, dev, 77, 1000, 0.4, n_steps, 5, 0.234,
                                          False, "f64", 0.0)
        if it == 0:
            assert eng.profile_report()["k_pcn_flow_fused"][0] == n_steps
            eng.profile(False)
        torch.testing.assert_close(lq, eng.coupling_logprob(x, dev), rtol=1e-5, atol=2e-3)
        torch.testing.assert_close(ll, eng.mixture_logpdf(x, t_ll), rtol=1e-9, atol=1e-9)
        if first is None:
            first = (x, lq, n_acc)
        else:
            assert torch.equal(x, first[0]) and torch.equal(lq, first[1]) and np.array_equal(n_acc, first[2]), it


@pytest.mark.parametrize("m,d,iters", [(2048, 32, 12), (4096, 64, 6), (16384, 128, 8), (2048, 32, 1), (1999, 32, 50)])
def test_student_fit_on_the_device_vs_numpy_em(eng, m, d, iters):
    """asmc_student_fit (every EM sweep on the stream: factorisation, E-step, weighted mean, the root in nu by a section search,
    scatter matrix; one synchronisation) against the all-numpy EM of student_t.fit_student_t on heavy-tailed data: same
    iteration count, location / scale matrix / degrees of freedom to the rounding of two different factorisations and root
    finders; the factor it leaves for the mutation is the Cholesky factor of the returned scale matrix."""
    from aspire_amd.student_t import fit_student_t

    g = np.random.default_rng(m + d)
    A = g.normal(size=(d, d)) / np.sqrt(d)
    L0 = np.linalg.cholesky(A @ A.T + 0.5 * np.eye(d))
    x = 0.3 + (g.normal(size=(m, d)) @ L0.T) / np.sqrt(g.chisquare(6.0, size=m) / 6.0)[:, None]
    xd = eng.asarray(x)
    (mu_d, L_d, Linv_d), nu, n_it, status, mu_h, cov_h = eng.student_fit(xd, iters, 1e-3, 20.0)
    assert status == 0 and 1 <= n_it <= iters
    # the numpy EM stopped by the same rule
    ref_mu, ref_cov, ref_nu = fit_student_t(x, max_iter=iters)
    np.testing.assert_allclose(mu_h, ref_mu, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(cov_h, ref_cov, rtol=1e-7, atol=1e-9)
    assert nu == pytest.approx(ref_nu, rel=1e-6)
    assert 2.0 < nu < 30.0 or iters == 1
    np.testing.assert_allclose(mu_d.cpu().numpy(), mu_h, rtol=0, atol=0)
    Lh = L_d.cpu().numpy()
    np.testing.assert_allclose(Lh @ Lh.T, cov_h, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(Linv_d.cpu().numpy() @ Lh, np.eye(d), atol=1e-10)


def test_sampler_with_a_maf_proposal_gpu(eng):
    """The reference's default flow class as the proposal at d = 8: the proposal draw in k_maf_sample and every mutation step in the
    one-kernel step (zero-padded to 32 dimensions inside the library) - no torch pass, no host round trip per step (round 3 ran
    MAFFlow's PyTorch modules here); evidence of the Gaussian product within its error bar."""
    from aspire_amd.flows import MAFFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    d, n = 8, 50000
    flow = MAFFlow(d, n_transforms=2, hidden_features=(32, 32), seed=3, device=eng.device)
    flow.fit(1.3 * np.random.default_rng(0).normal(size=(4000, d)), n_epochs=6)
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=flow, xp=torch, engine=eng, rng=np.random.default_rng(4))
    eng.profile(True)
    out = sp.sample(n, sampler_kwargs=dict(n_steps=8, step_fn="pcn"), store_sample_history=False)
    rep = eng.profile_report()
    eng.profile(False)
    assert "flow: device-side step loop" in sp.last_mutation_path
    assert rep["k_pcn_flow_fused"][0] == 8 * len(sp.history.beta) and rep["k_maf_sample"][0] >= 1 and "k_maf_logprob" not in rep
    assert abs(float(out.log_evidence) - 0.5 * d * math.log(math.pi)) < 5 * float(out.log_evidence_error) + 0.02
    assert 0.05 < np.mean(sp.history.mcmc_acceptance) < 0.99


def test_aspire_flow_preconditioning_gpu(eng):
    """`Aspire.sample_posterior(preconditioning="flow")` on the device engine (aspire.py:351-366): the chain runs in the latent
    space of a coupling flow refitted at every temperature, behind the probit stage of the prior box; bounded evidence within its
    error, samples inside the bounds."""
    from aspire_amd import Aspire
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.transforms import FlowPreconditioningTransform

    d, n = 4, 20000
    params = [f"x_{i}" for i in range(d)]

    def log_prior(s):
        x = torch.as_tensor(s.x)
        inside = (x.abs() <= 4.0).all(dim=1)
        return torch.where(inside, torch.full_like(x[:, 0], -d * math.log(8.0)), torch.full_like(x[:, 0], -float("inf")))

    def log_like(s):
        return -0.5 * (torch.as_tensor(s.x) ** 2).sum(1)

    true_logz = 0.5 * d * math.log(2 * math.pi) + d * math.log(math.erf(4 / math.sqrt(2))) - d * math.log(8.0)
    asp = Aspire(log_likelihood=log_like, log_prior=log_prior, dims=d, parameters=params, prior_bounds={p: (-4.0, 4.0) for p in params},
                 bounded_to_unbounded=True, xp=torch, flow=GaussianFlow(d, sigma=1.6, engine=eng, seed=3), hidden_features=(16, 16))
    out = asp.sample_posterior(n, sampler="smc", rng=np.random.default_rng(4), preconditioning="flow",
                               preconditioning_kwargs=dict(fit_kwargs=dict(n_epochs=2)), sampler_kwargs=dict(n_steps=4),
                               store_sample_history=False, engine=eng)
    assert isinstance(asp.sampler.preconditioning_transform, FlowPreconditioningTransform)
    assert asp.sampler.preconditioning_transform.flow is not None
    assert bool((torch.as_tensor(out.x).abs() <= 4.0).all())
    assert abs(float(out.log_evidence) - true_logz) < 5 * float(out.log_evidence_error) + 0.05, (float(out.log_evidence), true_logz)
