"""Restatement-to-reference speed ratio, measured in the BUILD container only.

Times the REAL reference (mj-will/aspire, imported through oracle/ref_shim.py) and the C oracle
(oracle/asmc_oracle.c, the CPU baseline `kind: "port"` that travels to the GPU box) on the same IS-only temperature
iteration of BASELINE configs[1] (1M x 32 fp64; smc/base.py:401-445 without mutate: determine_beta, ESS, evidence ratio
+ variance, resample).  The mutation step cannot be timed on the reference: its arithmetic lives in the absent
third-party `minipcn`.  Writes profiles/ref_ratio.json, which bench.py quotes next to its CPU baseline.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle as O
    import ref_shim
    from conftest import synth

    rs, smc, _, _ = ref_shim.import_reference()
    n, d = 1_000_000, 32
    x, ll, lp, lq = synth(n, d, 0)

    class Stub(smc.SMCSampler):
        def mutate(self, particles, beta, n_steps=None):
            return particles

    def ref_iteration(seed):
        sp = Stub(log_likelihood=None, log_prior=None, dims=d, prior_flow=None, xp=np, rng=np.random.default_rng(seed))
        sp.target_efficiency, sp.target_efficiency_rate, sp.adaptive, sp.adaptive_min_beta_step = 0.5, 1.0, True, False
        s = rs.SMCSamples(x=x, log_likelihood=ll, log_prior=lp, log_q=lq, beta=0.0)
        t0 = time.perf_counter()
        beta, _ = sp.determine_beta(s, 0.0, float("nan"), 0.0, max_beta_step=1.0, beta_tolerance=1e-6)
        ess = s.log_weights(beta)
        ratio, var = s.log_evidence_ratio(beta), s.log_evidence_ratio_variance(beta)
        out = s.resample(beta, rng=sp.rng)
        return time.perf_counter() - t0, beta, float(ratio), np.asarray(out.x)[:4].copy(), ess is not None, var

    def port_iteration(seed):
        st = O.pcg64_state_from_numpy(np.random.default_rng(seed))
        t0 = time.perf_counter()
        (xo, _, _, _), sc = O.is_iteration(x, ll, lp, lq, 0.0, 0.5, 1e-6, st)
        return time.perf_counter() - t0, float(sc[0]), xo[:4].copy()

    ref_iteration(1), port_iteration(1)  # warm
    tr, tp = [], []
    for k in range(3):
        a = ref_iteration(10 + k)
        b = port_iteration(10 + k)
        assert a[1] == b[1] and np.array_equal(a[3], b[2])  # same beta*, same resampled rows
        tr.append(a[0]), tp.append(b[0])
    res = {
        "workload": "configs[1] IS-only temperature iteration, 1M x 32 fp64 (determine_beta tol 1e-6 target 0.5, "
                    "evidence ratio + variance, multinomial resample)",
        "where": f"build container, {os.cpu_count()} cores, single thread each",
        "reference_s_per_iteration": min(tr), "port_s_per_iteration": min(tp),
        "reference_particle_iterations_per_s": n / min(tr), "port_particle_iterations_per_s": n / min(tp),
        "port_over_reference_speed": min(tr) / min(tp),
        "measured": "round 4 build container, re-run of tests/tools/ref_ratio.py on the round's oracle",
        "note": "identical beta* and resampled rows on both sides; the mutation step has no reference timing (minipcn absent)",
    }
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "ref_ratio.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
