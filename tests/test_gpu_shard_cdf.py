"""Sharded exact cdf (include/asmc.h asmc_cdf_shard_*) and the ordered range selection, with the ranks emulated in one
process: every shard's slice of numpy's cumsum over the GLOBAL weight vector, divided by the global total, bit for bit -
or an explicit failure flag, never a silently different value.  The two-process forms (collectives included) are
tests/test_dist_gloo.py (CPU double) and tests/test_gpu_dist.py (two ranks on one GPU)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _weights

pytestmark = pytest.mark.gpu


def _sharded_cdf(eng, w, bounds, approx_err=0.0):
    """Run the two entry points for every emulated rank; returns ([cdf slices], [edges {fail, total, lo, hi}])."""
    cuts = [0] + list(bounds) + [len(w)]
    shards = [eng.asarray(w[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    recs, cdfs = [], []
    for r, ws in enumerate(shards):
        carry = float(np.sum(w[:cuts[r]])) * (1.0 + approx_err)  # parallel-order sum: approximate on purpose
        cdf, rec = eng.cdf_shard_records(ws, carry, r == 0)
        recs.append(rec)
        cdfs.append(cdf)
    recs_all = torch.cat(recs, dim=0).contiguous()
    tiles = np.cumsum([0] + [rc.shape[0] for rc in recs])
    world = len(shards)
    r1 = [eng.cdf_shard_chain(ws, cdf, recs_all, int(tiles[r]), None, None, world, r) for r, (ws, cdf) in enumerate(zip(shards, cdfs))]
    states = torch.cat([st for _, st in r1]).contiguous()  # the all-gather between the rounds
    edges = []
    for r, (ws, cdf) in enumerate(zip(shards, cdfs)):
        work, st = eng.cdf_shard_chain(ws, cdf, recs_all, int(tiles[r]), r1[r][0], states, world, r)
        edges.append(eng.cdf_shard_finish(ws, cdf, recs_all, int(tiles[r]), work, st).cpu().numpy())
    return [c.cpu().numpy() for c in cdfs], edges, cuts


@pytest.mark.parametrize("kind", ["smooth", "heavy", "equal", "ties"])
@pytest.mark.parametrize("n,bounds", [(10000, [4097]), (300001, [100000, 200001]), (1 << 20, [1 << 19]),
                                       (1_000_003, [125000, 250001, 374999, 500000, 625003, 750000, 875000]),
                                       (4099, [1, 2049])])
def test_shard_cdf_is_the_global_numpy_cumsum_slice(hip_engine, n, bounds, kind):
    eng = hip_engine
    w = _weights(n, 23 + n % 1000, kind)
    ref = np.cumsum(w)
    ref = ref / ref[-1]
    slices, edges, cuts = _sharded_cdf(eng, w, bounds)
    if any(e[0] != 0.0 for e in edges):
        # the chain could not be closed in two rounds on some rank (failing tiles owned by several ranks that block each
        # other): the caller then repeats with the replicated scan on EVERY rank (tests/test_dist_gloo.py covers that path).
        # It only happens where tiles really cross several binades: heavy-tailed / dyadic weights, or a rank boundary a few
        # thousand particles into the population (the running sum still doubles several times per tile there) ...
        assert all(e[0] in (0.0, 1.0) for e in edges)
        assert kind in ("heavy", "ties") or bounds[0] < 8192
    # ... and a rank that reports success holds the right slice, whatever the others report
    for r, (got, e) in enumerate(zip(slices, edges)):
        if e[0] != 0.0:
            continue
        a, b = cuts[r], cuts[r + 1]
        assert np.array_equal(got, ref[a:b]), (r, np.flatnonzero(got != ref[a:b])[:5])
        assert e[1] == np.cumsum(w)[-1]  # exact global total
        assert e[2] == (ref[a - 1] if a > 0 else 0.0) and e[3] == ref[b - 1]  # the slice [lo, hi) of the global cdf
    assert edges[-1][0] != 0.0 or edges[-1][3] == 1.0


def test_shard_cdf_smooth_weights_never_need_the_fallback(hip_engine):
    """The production case (normalised importance weights at ESS = N/2, equal shards): the chain verifies every record."""
    from conftest import synth

    import oracle as O

    n = 1 << 20
    _, ll, lp, lq = synth(n, 8, 77)
    w = O.normalized_weights(ll, lp, lq, 0.0, 0.06)
    for world in (2, 8):
        bounds = [n * r // world for r in range(1, world)]
        slices, edges, cuts = _sharded_cdf(hip_engine, w, bounds)
        assert all(e[0] == 0.0 for e in edges)
        ref = np.cumsum(w)
        ref /= ref[-1]
        assert np.array_equal(np.concatenate(slices), ref)


def test_shard_cdf_wrong_hint_raises_the_flag_not_a_wrong_value(hip_engine):
    """An approximate carry that is off by a factor (another binade) must fail verification on every rank."""
    w = _weights(200000, 5, "smooth")
    slices, edges, cuts = _sharded_cdf(hip_engine, w, [70000, 140000], approx_err=1.5)
    assert all(e[0] == 1.0 for e in edges)
    # a hint that is only a few ulps off (what the sampler passes: a rank-ordered sum of partial sums) is fine
    slices, edges, cuts = _sharded_cdf(hip_engine, w, [70000, 140000], approx_err=3e-16)
    ref = np.cumsum(w)
    ref /= ref[-1]
    assert all(e[0] == 0.0 for e in edges) and np.array_equal(np.concatenate(slices), ref)


@pytest.mark.parametrize("n", [1, 63, 2049, 300001, 1 << 21])
def test_select_range_is_ordered_boolean_indexing(hip_engine, n):
    u = np.random.default_rng(n).random(n)
    for lo, hi in ((0.0, 1.0), (0.25, 0.5), (0.999, 1.0), (0.3, 0.3000001), (0.5, 0.5)):
        lohi = hip_engine.asarray(np.array([lo, hi]))
        got = hip_engine.select_range(hip_engine.asarray(u), lohi).cpu().numpy()
        assert np.array_equal(got, u[(u >= lo) & (u < hi)])


def test_owner_selection_reproduces_generator_choice_split_by_owner(hip_engine):
    """The whole owner-layout selection for 4 emulated ranks at 1M: rank r's kept draws searched in its slice give the
    sub-sequence of numpy's Generator.choice index vector that points into its shard, in draw order."""
    from conftest import synth

    import oracle as O
    from aspire_amd import smc_math

    eng, n, world = hip_engine, 1_000_000, 4
    _, ll, lp, lq = synth(n, 4, 91)
    w = O.normalized_weights(ll, lp, lq, 0.0, 0.05)
    ref = np.random.default_rng(12).choice(n, size=n, replace=True, p=w)
    bounds = [n * r // world for r in range(1, world)]
    cuts = [0] + bounds + [n]
    shards = [eng.asarray(w[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    recs, cdfs = [], []
    for r, ws in enumerate(shards):
        cdf, rec = eng.cdf_shard_records(ws, float(np.sum(w[:cuts[r]])), r == 0)
        recs.append(rec), cdfs.append(cdf)
    recs_all = torch.cat(recs).contiguous()
    t0 = np.cumsum([0] + [rc.shape[0] for rc in recs])
    u_all = smc_math.draw_uniforms(eng, np.random.default_rng(12), n, 0, n)
    total = 0
    r1 = [eng.cdf_shard_chain(shards[r], cdfs[r], recs_all, int(t0[r]), None, None, world, r) for r in range(world)]
    states = torch.cat([st for _, st in r1]).contiguous()
    for r in range(world):
        work, st = eng.cdf_shard_chain(shards[r], cdfs[r], recs_all, int(t0[r]), r1[r][0], states, world, r)
        edges = eng.cdf_shard_finish(shards[r], cdfs[r], recs_all, int(t0[r]), work, st)
        assert float(edges[0]) == 0.0
        kept = eng.select_range(u_all, edges[2:4])
        idx = eng.search(cdfs[r], kept).cpu().numpy() + cuts[r]
        assert np.array_equal(idx, ref[(ref >= cuts[r]) & (ref < cuts[r + 1])])
        total += idx.size
    assert total == n
