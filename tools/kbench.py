#!/usr/bin/env python
"""Kernel micro-benchmarks (HIP-event timed) for the hot-path ops.  Usage: python tools/kbench.py [op ...]
ops: pcn gram colsum cdf search gather weights flow"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ops = sys.argv[1:] or ["pcn", "gram", "colsum", "cdf", "search", "gather", "weights"]
    n, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 32))
    eng = HipEngine(0, n_max=n, d_max=max(d, 32))
    g = torch.Generator(device="cuda").manual_seed(0)
    if "flow" in ops:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from conftest import random_coupling_flow

        flow = random_coupling_flow(d, 4, 64)
        dev = flow.device_coupling(eng)
        flops = n * 4 * 2 * ((d // 2) * 64 + 64 * 64 + 64 * d)
        for xdt in (torch.float64, torch.float32):
            xx = torch.randn((n, d), device="cuda", dtype=xdt, generator=g)
            ms = timeit(lambda: eng.coupling_logprob(xx, dev))
            flow.to("cuda")
            mt = timeit(lambda: flow.log_prob(xx), reps=3, warm=1)
            tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
            a = eng.mixture_logpdf(xx, tgt)
            b, c = a.clone(), eng.coupling_logprob(xx, dev)
            mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
            eng.profile(True)
            mp = timeit(lambda: eng.pcn_mutate_flow(xx, a, b, c, 0.3, mu, eye, eye, tgt, tgt, dev, 7, 0, 0.3, 8, 0, 0.234, True), reps=3, warm=1) / 8
            rep = eng.profile_report()
            eng.profile(False)
            print(f"pcn+flow x={xdt}: {mp*1e3:.1f} us/step  " + "  ".join(f"{k}={v[1]*1e3:.0f}us" for k, v in rep.items()))
            print(f"flow x={xdt}: hip {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s fp32   torch modules {mt*1e3:.1f} us")
    for xdt, sb in ((torch.float64, 8), (torch.float32, 4)):
        x = (1.5 * torch.randn((n, d), device="cuda", dtype=torch.float64, generator=g)).to(xdt)
        tgt = eng.make_mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
        q = eng.make_mixture([-d * np.log(1.5) - 0.5 * d * np.log(2 * np.pi)], np.zeros((1, d)), np.full((1, d), 1 / 2.25))
        ll = eng.mixture_logpdf(x, tgt)
        lp, lq = ll.clone(), eng.mixture_logpdf(x, q)
        if "pcn" in ops:
            mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
            for noise in ("f64", "f32"):
                xs, a, b, c = x.clone(), ll.clone(), lp.clone(), lq.clone()
                ms = timeit(lambda: eng.pcn_mutate(xs, a, b, c, 0.3, mu, eye, eye, tgt, tgt, q, 7, 0, 0.3, 8, 0, 0.234, True, noise), reps=3, warm=1) / 8
                print(f"pcn x={xdt} noise={noise}: {ms*1e3:.1f} us/step  {(2*d*sb+16)*n/ms/1e6:.0f} GB/s alg")
        if xdt == torch.float32:
            continue
        if "gram" in ops:
            mean = eng.colsum(x) / n
            print(f"colsum: {timeit(lambda: eng.colsum(x))*1e3:.1f} us   gram: {timeit(lambda: eng.centered_gram(x, mean))*1e3:.1f} us")
        w = eng.normalized_weights(ll, lp, lq, 0.0, 0.05, 0.0, 0.0)
        w = w / w.sum()
        if "cdf" in ops:
            for mode in ("exact", "fast"):
                print(f"cdf {mode}: {timeit(lambda: eng.cdf(w, mode))*1e3:.1f} us")
        cdf, tot = eng.cdf(w, "fast")
        eng.cdf_normalize(cdf, tot)
        u = torch.rand(n, device="cuda", dtype=torch.float64, generator=g)
        if "search" in ops:
            print(f"search: {timeit(lambda: eng.search(cdf, u))*1e3:.1f} us")
        idx = eng.search(cdf, u)
        if "gather" in ops:
            ms = timeit(lambda: eng.gather(idx, x, ll, lp, lq))
            print(f"gather: {ms*1e3:.1f} us  {(2*(d*sb+24)+8)*n/ms/1e6:.0f} GB/s alg")
        if "weights" in ops:
            for K in (1, 2, 15):
                betas = np.linspace(0.01, 0.9, K)
                print(f"weights_stats K={K}: {timeit(lambda: eng.weights_stats(ll, lp, lq, 0.0, betas))*1e3:.1f} us (incl. host sync)")


if __name__ == "__main__":
    main()
