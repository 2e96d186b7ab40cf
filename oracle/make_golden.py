"""Generate the golden vectors in tests/golden/ from the REAL reference (container-only).

Run:  python oracle/make_golden.py
Needs /root/reference (imported through oracle/ref_shim.py).  Only data (seeded synthetic inputs
and the reference's outputs) is written; no reference source travels.

Files
  ref_weights.npz    G1  log_weights / ESS / log_evidence_ratio / variance   (samples.py:1221-1249, utils.py:248-255,510-512)
  ref_beta.npz       G2  SMCSampler.determine_beta                            (smc/base.py:123-213)
  ref_resample.npz   G3  SMCSamples.resample indices for default_rng(seed)    (samples.py:1251-1287)
  ref_loop.npz       G4  SMCSampler.sample history with deterministic stub mutate (smc/base.py:215-488)
  ref_initial.npz    G5  MCMCSampler.draw_initial_samples with invalid rows   (mcmc.py:49-110)
  ref_anchors.npz        the tiny known-answer cases of reference tests/test_samples.py:637-731
  ref_transforms.npz G6  CompositeTransform fit / forward / inverse + log|det J|     (transforms.py:142-316,411-646)
  ref_checkpoint.npz G7  checkpoint state dicts of the G4 loop (keys, types, beta, iteration, generator state, particles),
                         the HDF5 layout the reference writes for them and for SMCHistory.save
                         (samplers/base.py:158-252, smc/base.py:521-562, history.py:83-112, utils.py:733-872)

`python oracle/make_golden.py ref_checkpoint` regenerates only the named files.
"""
from __future__ import annotations

import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ONLY = {a.replace(".npz", "") for a in sys.argv[1:]}


class StateCollector:
    """Checkpoint callback (module level: the reference records the sample() kwargs in the state's config, and the
    state must stay picklable)."""

    def __init__(self):
        self.states = []

    def __call__(self, state):
        self.states.append(dict(state))


def save(name, arrays):
    if ONLY and name.replace(".npz", "") not in ONLY:
        return
    np.savez_compressed(os.path.join(OUT, name), **arrays)


def synth(n, d, seed, sigma_q=1.5):
    """The synthetic Gaussian batch of BASELINE.md §3 (config 2 shape): x = sigma_q*N(0,I),
    ll = lp = -0.5|x|^2, lq = log N(x; 0, sigma_q^2 I)."""
    g = np.random.default_rng(seed)
    x = sigma_q * g.normal(size=(n, d))
    ll = -0.5 * np.sum(x**2, axis=1)
    lp = ll.copy()
    lq = -0.5 * np.sum((x / sigma_q) ** 2, axis=1) - d * np.log(sigma_q) - 0.5 * d * np.log(2 * np.pi)
    return x, ll, lp, lq


def main():
    logging.disable(logging.CRITICAL)
    rs, smc, mcmc, ut = ref_shim.import_reference()
    os.makedirs(OUT, exist_ok=True)

    # ---------------- G1 weights -------------------------------------------------------------
    g1 = {}
    cases = []
    for n, d, seed in [(10, 2, 11), (2000, 4, 12), (65536, 8, 13)]:
        x, ll, lp, lq = synth(n, d, seed)
        for beta0 in (0.0, 0.3):
            for beta in (beta0 + 0.01, 0.5, 1.0):
                s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=beta0)
                lw = s.log_weights(beta)
                key = f"n{n}_b{beta0}_t{beta}"
                cases.append((n, d, seed, beta0, beta))
                g1[key + "_lse_unnorm"] = float(ut.logsumexp(s.unnormalized_log_weights(beta)))
                g1[key + "_ess"] = float(ut.effective_sample_size(lw))
                g1[key + "_ratio"] = float(s.log_evidence_ratio(beta))
                g1[key + "_var"] = float(s.log_evidence_ratio_variance(beta))
                if n <= 2000:
                    g1[key + "_lw"] = np.asarray(lw)
                else:
                    g1[key + "_lw_stride"] = np.asarray(lw)[::257]
                    g1[key + "_lw_sum"] = float(np.sum(lw))
    g1["cases"] = np.array(cases, dtype=np.float64)
    save("ref_weights.npz", g1)

    # ---------------- G2 determine_beta ------------------------------------------------------
    class _Flow:
        xp = np

    def mk_sampler(cls=smc.SMCSampler, **kw):
        return cls(log_likelihood=lambda s: None, log_prior=lambda s: None, dims=4, prior_flow=_Flow(), xp=np, **kw)

    g2 = {}
    cases = []
    for n, d, seed in [(2000, 4, 0), (65536, 8, 21)]:
        if seed == 0:  # the SURVEY anchor: X = 2*normal, lq = N(0,4I)
            x, ll, lp, lq = synth(n, d, 0, sigma_q=2.0)
        else:
            x, ll, lp, lq = synth(n, d, seed)
        for beta0 in (0.0, 0.2):
            for tol in (1e-6, 1e-8):
                for ti, target in enumerate([0.5, (0.3, 0.9)]):
                    sp = mk_sampler()
                    sp.adaptive = True
                    sp.adaptive_min_beta_step = False
                    sp.target_efficiency = target
                    sp.target_efficiency_rate = 1.0
                    s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=beta0)
                    b, ms = sp.determine_beta(s, beta0, np.nan, 0.0, max_beta_step=1.0, beta_tolerance=tol)
                    cases.append((n, d, seed, beta0, tol, ti, float(b)))
        # adaptive min step + max step clamp
        sp = mk_sampler()
        sp.adaptive = True
        sp.adaptive_min_beta_step = True
        sp.target_efficiency = 0.5
        sp.target_efficiency_rate = 1.0
        s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=0.0)
        b, ms = sp.determine_beta(s, 0.0, np.nan, 1 / 5, max_beta_step=0.25, beta_tolerance=1e-6)
        g2[f"n{n}_minstep"] = np.array([float(b), float(ms)])
    g2["cases"] = np.array(cases, dtype=np.float64)
    save("ref_beta.npz", g2)

    # ---------------- G3 resample indices ----------------------------------------------------
    g3 = {}
    cases = []
    for n, d, seed in [(10, 2, 31), (2000, 4, 32), (65536, 8, 33)]:
        x, ll, lp, lq = synth(n, d, seed)
        for beta0, beta in [(0.0, 0.05), (0.3, 0.6)]:
            for n_out in (n, 7, 2 * n):
                s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=beta0)
                rng = np.random.default_rng(1000 + seed)
                out = s.resample(beta, n_samples=n_out, rng=rng)
                # recover indices: rows are unique with probability 1 -> match on log_q + x[:,0]
                order = np.argsort(lq, kind="stable")
                pos = np.searchsorted(lq[order], out.log_q)
                idx = order[pos]
                assert np.array_equal(x[idx], out.x) and np.array_equal(ll[idx], out.log_likelihood)
                key = f"n{n}_b{beta0}_t{beta}_o{n_out}"
                g3[key + "_idx"] = idx.astype(np.int64)
                g3[key + "_next_u"] = rng.random(3)  # generator state after the call
                cases.append((n, d, seed, beta0, beta, n_out))
    # same-beta branch with n_samples != N (uniform weights, samples.py:1273-1274)
    x, ll, lp, lq = synth(2000, 4, 32)
    s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=0.4)
    rng = np.random.default_rng(77)
    out = s.resample(0.4, n_samples=50, rng=rng)
    order = np.argsort(lq, kind="stable")
    g3["samebeta_idx"] = order[np.searchsorted(lq[order], out.log_q)].astype(np.int64)
    g3["cases"] = np.array(cases, dtype=np.float64)
    save("ref_resample.npz", g3)

    # ---------------- G4 whole loop with stub mutate -----------------------------------------
    class GaussFlow:
        xp = np

        def __init__(self, d, sigma, seed):
            self.d, self.sigma, self.g = d, sigma, np.random.default_rng(seed)

        def log_prob(self, x):
            x = np.asarray(x)
            return (
                -0.5 * np.sum((x / self.sigma) ** 2, axis=1)
                - self.d * np.log(self.sigma)
                - 0.5 * self.d * np.log(2 * np.pi)
            )

        def sample_and_log_prob(self, n):
            x = self.sigma * self.g.normal(size=(n, self.d))
            return x, self.log_prob(x)

    def log_like(s):
        return -0.5 * np.sum(np.asarray(s.x) ** 2, axis=1)

    class StubSMC(smc.SMCSampler):
        kind = "identity"

        def mutate(self, particles, beta, n_steps=None):
            if self.kind == "identity":
                return particles
            # seeded random-walk Metropolis on the tempered target, 3 steps, host numpy
            x = np.array(particles.x, copy=True)
            for _ in range(3):
                lp_old = self.log_prob(x, beta)
                prop = x + 0.5 * self.rng.normal(size=x.shape)
                lp_new = self.log_prob(prop, beta)
                acc = np.log(self.rng.random(len(x))) < (lp_new - lp_old)
                x[acc] = prop[acc]
            out = rs.SMCSamples(x, xp=self.xp, beta=beta, dtype=self.dtype, parameters=self.parameters)
            out.log_q = self.prior_flow.log_prob(out.x)
            out.log_prior = self.log_prior(out)
            out.log_likelihood = self.log_likelihood(out)
            return out

    g4 = {}
    for name, kind, kwargs in [
        ("identity_adaptive", "identity", dict(adaptive=True, target_efficiency=0.5, beta_tolerance=1e-6)),
        ("rw_adaptive", "rw", dict(adaptive=True, target_efficiency=0.5, beta_tolerance=1e-6)),
        ("rw_adaptive_ramp", "rw", dict(adaptive=True, target_efficiency=(0.3, 0.9), beta_tolerance=1e-8)),
        ("rw_fixed20", "rw", dict(adaptive=False, n_steps=20)),
        ("rw_maxsteps", "rw", dict(adaptive=True, max_n_steps=4, target_efficiency=0.5)),
    ]:
        sp = StubSMC(
            log_likelihood=log_like, log_prior=log_like, dims=4, prior_flow=GaussFlow(4, 2.0, 5), xp=np,
            rng=np.random.default_rng(9),
        )
        sp.kind = kind
        sp.sampler_kwargs = {}
        out = sp.sample(2000, store_sample_history=False, **kwargs)
        h = sp.history
        g4[name + "_beta"] = np.array(h.beta)
        g4[name + "_ess"] = np.array(h.ess)
        g4[name + "_ess_target"] = np.array(h.ess_target)
        g4[name + "_eff_target"] = np.array(h.eff_target)
        g4[name + "_log_norm_ratio"] = np.array(h.log_norm_ratio)
        g4[name + "_log_norm_ratio_var"] = np.array(h.log_norm_ratio_var)
        g4[name + "_log_evidence"] = float(out.log_evidence)
        g4[name + "_log_evidence_error"] = float(out.log_evidence_error)
        g4[name + "_x_final"] = np.asarray(out.x)
        g4[name + "_nlike"] = sp.n_likelihood_evaluations
    save("ref_loop.npz", g4)


    # ---------------- G7 checkpoint state of the G4 loop + HDF5 layout --------------------------
    import json
    import pickle

    from fake_h5 import FakeGroup

    sp = StubSMC(log_likelihood=log_like, log_prior=log_like, dims=4, prior_flow=GaussFlow(4, 2.0, 5), xp=np,
                 rng=np.random.default_rng(9))
    sp.kind = "rw"
    sp.sampler_kwargs = {}
    collector = StateCollector()
    out = sp.sample(2000, store_sample_history=False, adaptive=True, target_efficiency=0.5, beta_tolerance=1e-6,
                    checkpoint_callback=collector, checkpoint_every=1)
    states = collector.states
    g7 = {"n_states": len(states), "final_beta": np.array(sp.history.beta), "final_x": np.asarray(out.x),
          "final_log_evidence": float(out.log_evidence)}
    for i, st in enumerate(states):
        g7[f"s{i}_keys"] = np.array(list(st.keys()))  # insertion order is part of what the reference writes
        g7[f"s{i}_types"] = np.array([type(v).__name__ for v in st.values()])
        g7[f"s{i}_iteration"] = int(st["iteration"])
        g7[f"s{i}_beta"] = float(st["meta"]["beta"])
        g7[f"s{i}_meta_keys"] = np.array(list(st["meta"].keys()))
        g7[f"s{i}_config"] = json.dumps(st["config"], sort_keys=True, default=str)
        g7[f"s{i}_sampler"] = st["sampler"]
        g7[f"s{i}_sampler_kwargs"] = json.dumps(st["sampler_kwargs"], sort_keys=True, default=str)
        rs_ = st["rng_state"]
        g7[f"s{i}_rng"] = json.dumps({"bit_generator": rs_["bit_generator"], "state": str(rs_["state"]["state"]),
                                      "inc": str(rs_["state"]["inc"]), "has_uint32": int(rs_["has_uint32"]),
                                      "uinteger": int(rs_["uinteger"])})
        h = st["history"]
        g7[f"s{i}_hist_beta"] = np.array(h.beta)
        g7[f"s{i}_hist_log_norm_ratio"] = np.array(h.log_norm_ratio)
        g7[f"s{i}_hist_log_norm_ratio_var"] = np.array(h.log_norm_ratio_var)
        g7[f"s{i}_hist_ess"] = np.array(h.ess)
        smp = st["samples"]
        g7[f"s{i}_samples_type"] = type(smp).__name__
        g7[f"s{i}_samples_dtypes"] = np.array([str(np.asarray(v).dtype) for v in (smp.x, smp.log_likelihood, smp.log_prior, smp.log_q)])
        g7[f"s{i}_samples_beta"] = float(smp.beta)
        if i in (1, len(states) - 1):  # full particle state at one mid-run checkpoint (resume test) and at the forced last one
            g7[f"s{i}_x"], g7[f"s{i}_ll"] = np.asarray(smp.x), np.asarray(smp.log_likelihood)
            g7[f"s{i}_lp"], g7[f"s{i}_lq"] = np.asarray(smp.log_prior), np.asarray(smp.log_q)
    # what the reference's own writers put into an HDF5 file (through the in-memory group protocol)
    f = FakeGroup()
    sp.save_checkpoint_to_hdf(states[1], f, path="checkpoint", dsetname="state")
    blob = f["checkpoint"]["state"]
    g7["h5_state_dtype"], g7["h5_state_ndim"] = str(blob.dtype), len(blob.shape)
    g7["h5_state_maxshape_none"] = int(blob.maxshape == (None,))
    assert pickle.loads(blob[...].tobytes())["iteration"] == states[1]["iteration"]
    sp.save_checkpoint_to_hdf(states[2], f, path="checkpoint", dsetname="state")  # overwrite in place (resize)
    assert pickle.loads(f["checkpoint"]["state"][...].tobytes())["iteration"] == states[2]["iteration"]
    f2 = FakeGroup()
    sp.history.save(f2, path="smc_history")
    lay = f2.layout()
    g7["h5_history_paths"] = np.array(sorted(lay))
    g7["h5_history_kinds"] = np.array([lay[k][0] for k in sorted(lay)])
    g7["h5_history_shapes"] = np.array([json.dumps(lay[k][1]) for k in sorted(lay)])
    g7["h5_history_beta"] = np.asarray(f2["smc_history"]["beta"][...])
    save("ref_checkpoint.npz", g7)

    # ---------------- G5 draw_initial_samples -------------------------------------------------
    class HoleFlow(GaussFlow):
        pass

    def lp_holes(s):
        x = np.asarray(s.x)
        v = -0.5 * np.sum(x**2, axis=1)
        return np.where(x[:, 0] > 1.0, -np.inf, v)

    def ll_holes(s):
        x = np.asarray(s.x)
        v = -0.5 * np.sum(x**2, axis=1)
        return np.where(x[:, 1] < -1.5, -np.inf, v)

    sp = mcmc.MCMCSampler(
        log_likelihood=ll_holes, log_prior=lp_holes, dims=3, prior_flow=HoleFlow(3, 1.0, 41), xp=np
    )
    init = sp.draw_initial_samples(500)
    g5 = dict(x=np.asarray(init.x), ll=np.asarray(init.log_likelihood), lp=np.asarray(init.log_prior),
              lq=np.asarray(init.log_q), nlike=sp.n_likelihood_evaluations)
    save("ref_initial.npz", g5)

    # ---------------- anchors from the reference's own tests ---------------------------------
    a = {}
    x = np.arange(20.0).reshape(10, 2)
    ll = np.linspace(0, 1, 10)
    s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=np.zeros(10), log_q=np.zeros(10), beta=0.2)
    a["t10_ratio"] = float(s.log_evidence_ratio(0.8))
    a["t10_ess"] = float(ut.effective_sample_size(s.log_weights(0.8)))
    a["t10_var"] = float(s.log_evidence_ratio_variance(0.8))
    a["t10_rows"] = s.resample(0.8, n_samples=7, rng=np.random.default_rng(42)).x[:, 0] / 2
    save("ref_anchors.npz", a)

    # ---------------- G6 preconditioning transforms -------------------------------------------
    import importlib

    tr = importlib.import_module("aspire.transforms")
    g6 = {}
    tcases = {
        # name: (bounds per dim (lo, hi) or None, periodic dims, bounded_to_unbounded, kind, affine)
        "mixed_logit_affine": ([(-1.0, 2.0), None, (0.0, 2 * np.pi)], [2], True, "logit", True),
        "probit": ([(-3.0, 5.0), (0.0, 1.0), (-10.0, 10.0), (2.0, 2.5)], [], True, "probit", False),
        "default_periodic": ([(-10.0, 10.0), (0.0, 1.0), (-1.0, 1.0), None, (0.0, 6.0)], [1, 4], False, "logit", False),
        "logit_affine_d32": ([(-10.0, 10.0)] * 32, [], True, "logit", True),
        "probit_affine_periodic": ([(-2.0, 2.0), (0.0, 3.0), None, (-1.0, 1.0)], [3], True, "probit", True),
    }
    for name, (bounds, per, b2u, kind, affine) in tcases.items():
        d = len(bounds)
        params = [f"p{j}" for j in range(d)]
        pb = {params[j]: (bounds[j] if bounds[j] is not None else (-np.inf, np.inf)) for j in range(d)}
        T = tr.CompositeTransform(parameters=params, periodic_parameters=[params[j] for j in per], prior_bounds=pb,
                                  bounded_to_unbounded=b2u, bounded_transform=kind, affine_transform=affine, xp=np,
                                  dtype=np.float64, eps=1e-6)
        g = np.random.default_rng(100 + d)
        cols = []
        for j in range(d):
            if bounds[j] is None:
                cols.append(2.0 * g.normal(size=120))
            elif j in per:  # periodic: also values outside the interval (they wrap)
                lo, hi = bounds[j]
                cols.append(g.uniform(lo - 1.5 * (hi - lo), hi + 1.5 * (hi - lo), size=120))
            else:
                lo, hi = bounds[j]
                u = g.uniform(size=120)
                u[:3] = [0.0, 1.0, 1e-9]  # on / next to the bounds: the eps clamp
                cols.append(lo + (hi - lo) * u)
        x = np.column_stack(cols)
        z_fit = T.fit(x)
        z, lj = T.forward(x)
        z2 = 1.5 * g.normal(size=(120, d))
        x2, lj2 = T.inverse(z2)
        g6[name + "_x"], g6[name + "_z_fit"], g6[name + "_z"], g6[name + "_lj"] = x, z_fit, z, lj
        g6[name + "_z2"], g6[name + "_x2"], g6[name + "_lj2"] = z2, x2, lj2
        g6[name + "_lower"] = np.array([pb[p][0] for p in params], dtype=np.float64)
        g6[name + "_upper"] = np.array([pb[p][1] for p in params], dtype=np.float64)
        g6[name + "_periodic"] = np.array([j in per for j in range(d)], dtype=np.int32)
        bounded = [b2u and bounds[j] is not None and j not in per for j in range(d)]
        g6[name + "_kind"] = np.array([({"logit": 1, "probit": 2}[kind] if b else 0) for b in bounded], dtype=np.int32)
        g6[name + "_affine"] = np.array(int(affine))
        if affine:
            g6[name + "_mean"], g6[name + "_std"] = np.asarray(T._affine_transform._mean), np.asarray(T._affine_transform._std)
    g6["names"] = np.array(list(tcases))
    save("ref_transforms.npz", g6)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
