"""Split propose/accept path with torch callables as densities (what a user with custom likelihoods runs), 1M x 32."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspire_amd.engine import HipEngine

n, d = int(os.environ.get("N", 1_000_000)), 32
eng = HipEngine(0, n_max=n, d_max=32)
g = torch.Generator("cuda").manual_seed(0)
x = torch.randn((n, d), device="cuda", dtype=torch.float64, generator=g)
mu, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
f = lambda t: -0.5 * (t * t).sum(1)  # noqa: E731
ll, lp, lq = f(x), f(x), f(x / 1.5)
for nu in (0.0, 6.0):
    for _ in range(3):
        xp, q0, q1 = eng.pcn_propose(x, mu, eye, eye, 0.3, 5, 0, 1, nu=nu)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(20):
        xp, q0, q1 = eng.pcn_propose(x, mu, eye, eye, 0.3, 5, 0, t, nu=nu)
    torch.cuda.synchronize()
    tp = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for t in range(20):
        xp, q0, q1 = eng.pcn_propose(x, mu, eye, eye, 0.3, 5, 0, t, nu=nu)
        eng.pcn_accept(x, xp, ll, lp, lq, f(xp), f(xp), f(xp / 1.5), q0, q1, 0.5, 5, 0, t)
    torch.cuda.synchronize()
    print(f"nu={nu}: propose {tp*1e3:.3f} ms, full split step {(time.perf_counter()-t0)/20*1e3:.3f} ms")
eng.profile(True)
for t in range(10):
    xp, q0, q1 = eng.pcn_propose(x, mu, eye, eye, 0.3, 5, 0, 100 + t, nu=0.0)
    eng.pcn_accept(x, xp, ll, lp, lq, f(xp), f(xp), f(xp / 1.5), q0, q1, 0.5, 5, 0, 100 + t)
rep = eng.profile_report(); eng.profile(False)
print({k: (v[0], round(v[1] * 1e3, 1)) for k, v in rep.items()})

# the two device-resident sessions (no host round trip per step): x-state vs whitened-state
for nu in (0.0, 6.0):
    for kind in ("x", "y"):
        xs, l3 = x.clone(), [ll.clone(), lp.clone(), lq.clone()]
        for rep_i in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if kind == "x":
                eng.pcn_split_begin(0.3)
                for t in range(20):
                    xp, q0, q1 = eng.pcn_propose(xs, mu, eye, eye, 0.0, 5, 0, t, nu=nu)
                    eng.pcn_accept(xs, xp, *l3, f(xp), f(xp), f(xp / 1.5), q0, q1, 0.5, 5, 0, t, want_count=False)
                    eng.pcn_split_adapt(n, 0.234, t, True)
                na, _, rho = eng.pcn_split_end(20)
            else:
                sess = eng.pcn_ysplit_begin(xs, 0.5, mu, eye, eye, 5, 0, 0.3, 0.234, True, nu, os.environ.get("NOISE", "f64"))
                for t in range(20):
                    xp = eng.pcn_ysplit_propose(sess, t)
                    eng.pcn_ysplit_accept(sess, t, *l3, f(xp), f(xp), f(xp / 1.5), n, t)
                na, _, rho = eng.pcn_ysplit_end(sess, 20)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
        print(f"nu={nu} {kind}-state session: {dt*1e3:.3f} ms/step, mean accept {na.mean()/n:.3f}, rho {rho:.4f}")
eng.profile(True)
sess = eng.pcn_ysplit_begin(x, 0.5, mu, eye, eye, 5, 0, 0.3, 0.234, True, 0.0, os.environ.get("NOISE", "f64"))
for t in range(10):
    xp = eng.pcn_ysplit_propose(sess, t)
    eng.pcn_ysplit_accept(sess, t, ll, lp, lq, f(xp), f(xp), f(xp / 1.5), n, t)
eng.pcn_ysplit_end(sess, 10)
rep = eng.profile_report(); eng.profile(False)
print({k: (v[0], round(v[1] * 1e3, 1)) for k, v in rep.items()})
