#!/usr/bin/env python
"""Fuzz of the exact cumulative sum (asmc_cdf mode "exact" and the importance step's chain) against numpy.cumsum, bit for bit:
random lengths, weight laws that put binade crossings everywhere (geometric growth, scale steps, dominant weights, zeros,
subnormals, dyadic ties), with and without a carry.  ROUNDS, SEED env."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine  # noqa: E402

rounds, seed = int(os.environ.get("ROUNDS", 400)), int(os.environ.get("SEED", 0))
g = np.random.default_rng(seed)
eng = HipEngine(0, n_max=1 << 21, d_max=1)


def draw(n):
    kind = g.integers(0, 9)
    w = np.exp(g.normal(size=n) * g.choice([0.1, 1.0, 5.0, 40.0]))
    if kind == 1:  # growth: every add crosses for a while
        k = min(n, int(g.integers(1, 400)))
        w[:k] = np.ldexp(1.0 + g.random(k), (int(g.integers(1, 5)) * np.arange(k) - int(g.integers(0, 900))).clip(-1070, 900))
    elif kind == 2:  # scale steps at random places
        for _ in range(int(g.integers(1, 8))):
            w[int(g.integers(0, n)):] *= 2.0 ** int(g.integers(1, 40))
    elif kind == 3:  # zeros and a few dominant weights
        w[g.random(n) < 0.5] = 0.0
        w[g.integers(0, n, max(1, n // 500))] *= 2.0 ** 30
    elif kind == 4:  # dyadic ties
        w = np.ldexp(1.0, -g.integers(1, 60, n).astype(np.int64)).astype(np.float64)
    elif kind == 5:  # subnormal start
        k = min(n, 50)
        w[:k] = np.ldexp(g.random(k), -1074 + g.integers(0, 60, k))
    elif kind == 6:  # equal weights (sum ends at a binade edge after normalisation)
        w = np.full(n, 1.0 / n)
    elif kind == 7:  # normalised (the resampling case)
        w = w / w.sum()
    return w


bad = 0
for r in range(rounds):
    n = int(g.choice([g.integers(1, 70), g.integers(1, 5000), g.integers(2048, 300000), g.integers(1 << 20, (1 << 20) + 5000)],
                     p=[0.2, 0.4, 0.35, 0.05]))
    w = draw(n)
    carry = float(g.choice([0.0, 0.0, g.random(), np.ldexp(g.random(), int(g.integers(-300, 300)))]))
    with np.errstate(over="ignore"):
        ref = np.cumsum(np.concatenate([[carry], w]))[1:] if carry else np.cumsum(w)
    if not np.all(np.isfinite(ref)):
        continue
    got, total = eng.cdf(eng.asarray(w), "exact", carry)
    got = got.cpu().numpy()
    if not (np.array_equal(got, ref) and total == ref[-1]):
        bad += 1
        print("MISMATCH round", r, "n", n, "carry", carry, "first", np.flatnonzero(got != ref)[:3])
print(f"{rounds} rounds, {bad} mismatches")
sys.exit(1 if bad else 0)
