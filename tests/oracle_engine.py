"""OracleEngine — a TEST DOUBLE with HipEngine's method set, backed by the CPU oracle.

Lives under tests/ on purpose: the product (`aspire_amd/`) has no CPU path.  It lets the host
logic (smc_math, the sampler loop, sharding over gloo) run in the `-m "not gpu"` suite, following
the reference's own fake-backend test pattern (tests/test_samplers/test_mcmc/test_checkpointing.py:88-109).
Tensors are CPU torch tensors; every computation goes through oracle/oracle.py.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as O  # noqa: E402


class OracleMixture:
    def __init__(self, logw, mu, prec):
        self.mix = O.Mixture(logw, mu, prec)
        self.logw, self.mu, self.prec = self.mix.logw, self.mix.mu, self.mix.prec


def _np(t):
    return t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


class OracleEngine:
    name = "oracle"
    device = torch.device("cpu")
    search_rounds_cap = 16  # plain 16-ary rounds only (no prediction windows): the chain enqueues as many as the tolerance needs

    # ---- plumbing ---------------------------------------------------------------------------
    def ensure_capacity(self, n, d):
        pass

    def asarray(self, a, dtype=torch.float64):
        if isinstance(a, torch.Tensor):
            return a.to(dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)

    def empty(self, shape, dtype=torch.float64):
        return torch.empty(shape, dtype=dtype)

    def full(self, n, value):
        return torch.full((n,), value, dtype=torch.float64)

    def to_numpy(self, t):
        return _np(t)

    def synchronize(self):
        pass

    # ---- weighting --------------------------------------------------------------------------
    def weights_max(self, ll, lp, lq, beta0, betas):
        m, nn = [], 0
        for b in np.atleast_1d(betas):
            lw = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, float(b))
            nn += int(np.isnan(lw).sum())
            m.append(np.nanmax(lw) if not np.all(np.isnan(lw)) else -np.inf)
        return np.array(m), nn

    def weights_sums(self, ll, lp, lq, beta0, betas, m, shift=None):
        out = []
        for k, b in enumerate(np.atleast_1d(betas)):
            lw = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, float(b))
            sh = 0.0 if shift is None else float(np.atleast_1d(shift)[k])
            with np.errstate(all="ignore"):
                e = np.exp((lw + sh) - float(np.atleast_1d(m)[k]))
            out.append([np.sum(e), np.sum(e * e)])
        return np.array(out)

    def weights_stats(self, ll, lp, lq, beta0, betas):
        m, nn = self.weights_max(ll, lp, lq, beta0, betas)
        s = self.weights_sums(ll, lp, lq, beta0, betas, m)
        return np.column_stack([m, s[:, 0], s[:, 1], np.full(len(m), float(nn))])

    # sharded beta search: numpy restatement of asmc_find_beta_shard_{reduce,decide,result} (include/asmc.h)
    def find_beta_shard_reduce(self, ll, lp, lq, beta0, rnd, rec):
        from aspire_amd.smc_math import _bisection_tree

        st = self.__dict__.setdefault("_bis", {})
        a = (_np(ll) + _np(lp)) - _np(lq)  # Delta_i: lw_i(beta) = (beta - beta0) Delta_i up to rounding
        if rnd == 0:
            lw1 = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, 1.0)
            base, n_nan = float(np.nanmax(lw1)), float(np.isnan(lw1).sum())
            lo, hi = beta0, 1.0
        else:
            if st["done"]:
                return
            base, n_nan, lo, hi = st["m_one"], 0.0, st["lo"], st["hi"]
        betas = _bisection_tree(lo, hi, 4) + [hi]
        out = np.zeros(40)
        for k, b in enumerate(betas):
            lw = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, float(b))
            with np.errstate(all="ignore"):
                e = np.exp(lw - base * ((b - beta0) / (1.0 - beta0)))
            out[2 * k], out[2 * k + 1] = np.sum(e), np.sum(e * e)
        out[32], out[33] = base, n_nan
        rec[:] = torch.from_numpy(out)
        del a

    def find_beta_shard_decide(self, recs, world, n_global, beta0, target_eff, tol, rnd):
        from aspire_amd.smc_math import Stats, _bisection_tree, ess

        st = self.__dict__.setdefault("_bis", {})
        if rnd > 0 and st["done"]:
            return
        r = _np(recs).reshape(world, 40)
        m_all = float(r[:, 32].max())
        if rnd == 0:
            st.update(lo=beta0, hi=1.0, done=False, rounds=0, m_one=m_all, n_nan=float(r[:, 33].sum()), trip=None)
        lo, hi = st["lo"], st["hi"]
        betas = _bisection_tree(lo, hi, 4) + [hi]
        S = np.zeros(32)
        for k, b in enumerate(betas):
            t = (b - beta0) / (1.0 - beta0)
            for q in range(world):
                f = np.exp(r[q, 32] * t - m_all * t) if rnd == 0 else 1.0
                S[2 * k] += r[q, 2 * k] * f
                S[2 * k + 1] += r[q, 2 * k + 1] * f * f
        shift = [m_all * ((b - beta0) / (1.0 - beta0)) for b in betas]
        eff = [ess(Stats(shift[k], S[2 * k], S[2 * k + 1], n_global)) / n_global for k in range(16)]
        st["rounds"] += 1
        at_one = False
        if rnd == 0:
            st["eff_one"], st["one"] = eff[15], (m_all, S[30], S[31])
            if eff[15] >= target_eff:
                lo, at_one, st["trip"] = 1.0, True, (m_all, S[30], S[31])
        if not at_one:
            i = 0
            for _ in range(4):
                if not (hi - lo > tol):
                    break
                if eff[i] >= target_eff:
                    lo, st["trip"] = betas[i], (shift[i], S[2 * i], S[2 * i + 1])
                    i = 2 * i + 2
                else:
                    hi = betas[i]
                    i = 2 * i + 1
        st["lo"], st["hi"] = lo, hi
        st["done"] = not (hi - lo > tol)

    def find_beta_shard_result(self):
        st = self._bis
        return (float(st["lo"]), float(st["eff_one"]), bool(st["done"]), int(st["rounds"]), int(st["n_nan"]), st["trip"],
                tuple(map(float, st["one"])))

    # ---- the sharded importance step as one chain (HipEngine's shard-step entry points; host-logic double) -----------------
    # The double computes eagerly, so "the scalars stay on the device" means: they stay in self._bis / the tensors handed on,
    # and the host logic (smc_math.shard_step_enqueue / _wait / _check, SMCSamples.speculate_importance_step) reads them only
    # through shard_step_result - the same call order, the same collectives, the same decisions as on the GPU.
    def all_gather(self, comm, t, out=None):
        g = comm.all_gather_tensor(t.contiguous())
        if out is None:
            return g
        out.copy_(g.reshape(out.shape))
        return out

    def find_beta_shard_round(self, ll, lp, lq, beta0, target_eff, tol, world, n_global, rnd, recs_prev, rec):
        if rnd > 0:  # a round closes the previous one itself (asmc_find_beta_shard_round)
            self.find_beta_shard_decide(recs_prev, world, n_global, beta0, target_eff, tol, rnd - 1)
        self.find_beta_shard_reduce(ll, lp, lq, beta0, rnd, rec)

    def _shard_scalars(self, n_global):
        b, _, conv, _, n_nan, trip, _ = self.find_beta_shard_result()
        found = bool(conv and trip is not None and n_nan == 0 and b > self._bis_beta0)
        if not found:
            return None
        m, S1 = float(trip[0]), float(trip[1])
        shift = float((m + np.log(S1)) - np.log(float(n_global)))
        return b, m, S1 / n_global, shift, m + shift

    def weights_m2_lse_shard(self, ll, lp, lq, out, recs_last=None, world=1, n_global=0, beta0=0.0, target_eff=0.5, tol=1e-6,
                             n_rounds=0):
        if recs_last is not None:  # closes the search's last round (asmc_weights_m2_lse_shard)
            self.find_beta_shard_decide(recs_last, world, n_global, beta0, target_eff, tol, n_rounds - 1)
        self._bis_beta0, self._bis_n = float(beta0), int(n_global)
        sc = self._shard_scalars(n_global)
        m2 = s1p = 0.0
        if sc is not None:
            b, m, mean_u, shift, mp = sc
            m2, s1p = self.weights_m2_lse(ll, lp, lq, beta0, b, m, mean_u, shift, mp)
        out[0], out[1] = m2, s1p

    def normalized_weights_shard(self, ll, lp, lq, parts, world, rank, carry_uniform, state_copy=None):
        pr = _np(parts).reshape(world, 2)
        sc = self._shard_scalars(self._bis_n)
        s1p = float(pr[0, 1])
        below = 0.0
        for r in range(1, world):  # rank order (k_weights_map_shard)
            if r == rank:
                below = s1p
            s1p += float(pr[r, 1])
        found = sc is not None and s1p > 0.0 and np.isfinite(s1p)
        if found:
            b, _, _, shift, mp = sc
            w = self.normalized_weights(ll, lp, lq, self._bis_beta0, b, shift, float(mp + np.log(s1p)))
        else:
            w = torch.full((ll.numel(),), 1.0 / self._bis_n, dtype=torch.float64)
        self.rec_token = 0
        return w, torch.tensor([below / s1p if found else carry_uniform], dtype=torch.float64), None

    def select_range_dev(self, u, edges):
        kept = self.select_range(u, edges[2:4])
        buf = torch.empty_like(u)
        buf[: kept.numel()] = kept
        return buf, torch.tensor([kept.numel(), int(round(float(edges[0])))], dtype=torch.int64)

    def shard_step_result(self, res, world):
        r = _np(res)
        info = r[40 + 2 * world:].view(np.int64).reshape(world, 2).copy()
        return self.find_beta_shard_result(), r[40:40 + 2 * world].reshape(world, 2).copy(), info

    # Student-t reference fit: numpy restatement of asmc_student_estep / asmc_student_scale
    def student_estep(self, xs, mu, linv, nu):
        x = _np(xs).astype(np.float64)
        d = x.shape[1]
        y = (x - np.asarray(mu)) @ np.tril(np.asarray(linv)).T
        z = (nu + d) / (nu + (y * y).sum(1))
        return torch.from_numpy(z), float(z.sum()), float((np.log(z) - z).sum()), (z[:, None] * x).sum(0)

    def student_scale(self, xs, z, mu):
        return torch.from_numpy(np.sqrt(_np(z))[:, None] * (_np(xs).astype(np.float64) - np.asarray(mu)))

    def pcg64_select(self, state4, n_total, lo, hi):
        return torch.from_numpy(O.pcg64_select(np.array(state4, dtype=np.uint64), n_total, lo, hi))

    def weights_m2(self, ll, lp, lq, beta0, beta, m, mean_u):
        lw = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, beta)
        return float(np.sum((np.exp(lw - m) - mean_u) ** 2))

    def weights_m2_lse(self, ll, lp, lq, beta0, beta, m, mean_u, shift, mp):
        lw = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, beta)
        with np.errstate(all="ignore"):
            return float(np.sum((np.exp(lw - m) - mean_u) ** 2)), float(np.sum(np.exp((lw + shift) - mp)))

    def log_weights(self, ll, lp, lq, beta0, beta, shift):
        return torch.from_numpy(O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, beta) + shift)

    def normalized_weights(self, ll, lp, lq, beta0, beta, shift, lse):
        lw = O.unnormalized_log_weights(_np(ll), _np(lp), _np(lq), beta0, beta) + shift
        return torch.from_numpy(np.exp(lw - lse))

    def count_nonfinite(self, v):
        a = _np(v)
        return int(np.isnan(a).sum()), int(np.isinf(a).sum())

    # ---- the importance step as one call (HipEngine.importance_step / importance_result; host-logic double) ----------
    def importance_step(self, ll, lp, lq, beta0, target_eff, tol, state4, n_out):
        """Search (the sharded rounds over a one-rank world), evidence moments, exact resampling indices for the next
        n_out doubles of the PCG64 stream: what asmc_importance_step enqueues, computed eagerly."""
        from aspire_amd import _lib
        from aspire_amd.smc_math import Stats

        n = ll.numel()
        rec = torch.zeros(_lib.ASMC_BIS_REC, dtype=torch.float64)
        for rnd in range(64):
            self.find_beta_shard_reduce(ll, lp, lq, beta0, rnd, rec)
            self.find_beta_shard_decide(rec, 1, n, beta0, target_eff, tol, rnd)
            if self._bis["done"]:
                break
        b, eff1, conv, rounds, n_nan, trip, one = self.find_beta_shard_result()
        found = bool(conv and trip is not None and n_nan == 0 and b > beta0)
        m2 = s1p = 0.0
        if found:
            st = Stats(*trip, n)
            shift = float((st.m + np.log(st.S1)) - np.log(float(n)))
            mp = st.m + shift
            m2, s1p = self.weights_m2_lse(ll, lp, lq, beta0, b, st.m, st.S1 / n, shift, mp)
            w = self.normalized_weights(ll, lp, lq, beta0, b, shift, float(mp + np.log(s1p)))
        else:
            w = torch.full((n,), 1.0 / n, dtype=torch.float64)
        cdf, last = self.cdf(w, "exact", 0.0)
        cdf = self.cdf_normalize(cdf, last)
        idx = self.search(cdf, self.uniforms_pcg64(state4, 0, n_out))
        self._is_result = (b, eff1, conv, rounds, n_nan, trip, one, m2, s1p, found)
        return idx

    def importance_result(self):
        return self._is_result

    # ---- resampling -------------------------------------------------------------------------
    def cdf(self, w, mode="exact", carry_in=0.0):
        a = _np(w).copy()
        if mode == "exact":  # strictly sequential accumulation (numpy cumsum), with carry in front
            if carry_in != 0.0:
                out = np.cumsum(np.concatenate([[carry_in], a]))[1:]
            else:
                out = np.cumsum(a)
        else:
            out = carry_in + np.cumsum(a)
        return torch.from_numpy(out), float(out[-1])

    def cdf_normalize(self, cdf, last):
        cdf /= last
        return cdf

    def select_range(self, u, lohi):
        """u[(lo <= u) & (u < hi)] in index order (asmc_select_range)."""
        a, lo, hi = _np(u), float(lohi[0]), float(lohi[1])
        return torch.from_numpy(a[(a >= lo) & (a < hi)].copy())

    def uniforms_pcg64(self, state4, offset, n):
        st = np.array(state4, dtype=np.uint64)
        O.pcg64_advance(st, int(offset))
        return torch.from_numpy(O.pcg64_random(st, n))

    def systematic_uniforms(self, n_out, j0, n_total, u0, v=None):
        j = np.arange(j0, j0 + n_out, dtype=np.float64)
        off = u0 if v is None else _np(v)
        return torch.from_numpy((j + off) / float(n_total))

    def search(self, cdf, u):
        return torch.from_numpy(O.searchsorted_right(_np(cdf), _np(u)))

    def gather(self, idx, x, ll, lp, lq):
        xo, a, b, c = O.gather_rows(_np(idx), _np(x), _np(ll), _np(lp), _np(lq))
        return torch.from_numpy(xo), torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c)

    # ---- proposal / densities ---------------------------------------------------------------
    def make_mixture(self, logw, mu, prec):
        return OracleMixture(logw, mu, prec)

    def gaussian_draw(self, n, d, x_dtype, mu, sigma, seed, gid0, draw_id, want_lq=True):
        x = np.empty((n, d))
        for i in range(n):
            xi, _ = O.pcn_noise(seed, gid0 + i, draw_id, d)
            x[i] = _np(mu) + _np(sigma) * xi
        xt = torch.from_numpy(x).to(x_dtype)
        lq = None
        if want_lq:
            z = (xt.double().numpy() - _np(mu)) / _np(sigma)
            lq = torch.from_numpy(-0.5 * (z * z).sum(1) - np.log(_np(sigma)).sum() - 0.5 * d * np.log(2 * np.pi))
        return xt, lq

    def mixture_logpdf(self, x, mix):
        return torch.from_numpy(mix.mix.logpdf(_np(x).astype(np.float64)))

    def make_transform(self, kind, periodic, lower, upper, mean=None, std=None, eps=1e-6, unit_logj=0.0, affine_logj=0.0):
        return dict(kind=np.asarray(kind, dtype=np.int32), periodic=np.asarray(periodic, dtype=np.int32),
                    lower=np.asarray(lower, dtype=np.float64), upper=np.asarray(upper, dtype=np.float64),
                    mean=None if mean is None else np.asarray(mean, dtype=np.float64),
                    std=None if std is None else np.asarray(std, dtype=np.float64), eps=eps)

    def _transform(self, x, t, inverse, want_logj):
        y, lj = O.transform(_np(x).astype(np.float64), t["kind"], t["periodic"], t["lower"], t["upper"], t["mean"], t["std"],
                            t["eps"], inverse=inverse)
        return torch.from_numpy(y).to(x.dtype), (torch.from_numpy(lj) if want_logj else None)

    def transform_forward(self, x, t, want_logj=True):
        return self._transform(x, t, False, want_logj)

    def transform_inverse(self, z, t, want_logj=True):
        return self._transform(z, t, True, want_logj)

    def compact_valid(self, x, ll, lp, lq):
        # (float32 rows, as the HIP engine takes them: the oracle's row compaction is dtype blind - through float64 and back, exact)
        out = tuple(torch.from_numpy(a) for a in O.compact_valid(_np(x).astype(np.float64), _np(ll), _np(lp), _np(lq)))
        return (out[0].to(x.dtype),) + out[1:]

    # ---- moments / pCN ----------------------------------------------------------------------
    def colsum(self, x):
        return _np(x).astype(np.float64).sum(axis=0)

    def centered_gram(self, x, center):
        c = _np(x).astype(np.float64) - np.asarray(center)
        return c.T @ c

    def pcn_mutate(self, x, ll, lp, lq, beta, mu, L, Linv, t_ll, t_lp, t_lq, seed, gid0, rho, n_steps, step0=0,
                   target_accept=0.234, adapt=True, noise="f64", nu=0.0):
        assert x.dtype == torch.float64
        n = x.shape[0]
        n_acc, hist = np.zeros(n_steps, dtype=np.int64), np.zeros(n_steps)
        hook = getattr(self, "_count_hook", None)  # sharded: counts are global and the rate is over the whole population
        lag = int(adapt) if int(adapt) >= 2 else 1  # asmc_pcn_params.adapt = k >= 2: the block's updates at its end, in order
        pending = []
        self.count_exchanges = getattr(self, "count_exchanges", 0)
        for t in range(n_steps):
            hist[t] = rho
            if nu > 0.0:
                c = O.tpcn_step(_np(x), _np(ll), _np(lp), _np(lq), beta, _np(mu), _np(L), _np(Linv), rho, nu,
                                t_ll.mix, t_lp.mix, t_lq.mix, seed, gid0, step0 + t, noise)
            else:
                c = O.pcn_step(_np(x), _np(ll), _np(lp), _np(lq), beta, _np(mu), _np(L), _np(Linv), rho,
                               t_ll.mix, t_lp.mix, t_lq.mix, seed, gid0, step0 + t, noise)
            pending.append((t, int(c)))
            if lag > 1 and not ((t + 1) % lag == 0 or t == n_steps - 1):
                continue
            n_tot = n
            counts = np.array([float(v) for _, v in pending])
            if hook is not None:  # ONE exchange for the block's counts
                counts, n_tot = hook[0].all_gather_f64(counts).sum(axis=0), hook[1]
                self.count_exchanges += 1
            for (tp, _), cg in zip(pending, counts):
                n_acc[tp] = int(cg)
                if adapt:
                    rho = O.pcn_adapt(rho, int(cg) / n_tot, target_accept, tp)
            pending = []
        return n_acc, hist, rho

    # split path with device-resident step size / counts: host-side restatement of asmc_pcn_split_{begin,adapt,end}
    def pcn_split_begin(self, rho):
        self._split = {"rho": float(rho), "counts": [], "hist": [], "last": 0}

    def pcn_split_adapt(self, n_global, target_accept, t, adapt=True):
        sp = self._split
        c = sp["last"]
        hook = getattr(self, "_count_hook", None)
        if hook is not None:  # sharded: the global count (the product all-reduces a device cell through the exchange hook)
            c = int(hook[0].all_gather_f64(np.array([float(c)])).sum())
            n_global = hook[1]
        sp["counts"].append(c)
        sp["hist"].append(sp["rho"])
        if adapt:
            sp["rho"] = O.pcn_adapt(sp["rho"], c / n_global, target_accept, t)

    def pcn_split_end(self, n_steps):
        sp = self._split
        assert len(sp["counts"]) == n_steps
        return np.array(sp["counts"], dtype=np.int64), np.array(sp["hist"]), sp["rho"]

    def set_count_hook(self, comm, n_global):
        self._count_hook = None if (comm is None or n_global is None or comm.world == 1) else (comm, int(n_global))

    def pcn_propose(self, x, mu, L, Linv, rho, seed, gid0, step, nu=0.0):
        if rho == 0.0:  # device-resident step size of the split session
            rho = self._split["rho"]
        xn, mun, Ln, Li = _np(x).astype(np.float64), _np(mu), _np(L), _np(Linv)
        n, d = xn.shape
        y = (xn - mun) @ Li.T
        xi = np.stack([O.pcn_noise(seed, gid0 + i, step, d)[0] for i in range(n)])
        q0 = (y * y).sum(1)
        rs = np.full(n, rho)
        if nu > 0.0:
            g = np.array([O.gamma_unit(0.5 * (d + nu), seed, gid0 + i, step) for i in range(n)])
            rs = rho * np.sqrt((nu + q0) / (2.0 * g))
        yp = np.sqrt(1 - rho * rho) * y + rs[:, None] * xi
        q1 = (yp * yp).sum(1)
        xp = (mun + yp @ Ln.T).astype(xn.dtype if x.dtype == torch.float64 else np.float32)
        if nu > 0.0:  # the accept step adds half of these: twice the Student-t correction
            q0, q1 = 2.0 * O.tpcn_corr(q0, d, nu), 2.0 * O.tpcn_corr(q1, d, nu)
        return torch.from_numpy(xp).to(x.dtype), torch.from_numpy(q0), torch.from_numpy(q1)

    # whitened-state session of the split path: host-side restatement of asmc_pcn_ysplit_{begin,propose,accept,end}
    def pcn_ysplit_begin(self, x, beta, mu, L, Linv, seed, gid0, rho, target_accept=0.234, adapt=True, nu=0.0, noise="f64"):
        n, d = x.shape
        if d not in (4, 8, 16, 32):
            return None
        self.pcn_split_begin(rho)
        npdt = np.float64 if x.dtype == torch.float64 else np.float32
        y = ((_np(x).astype(np.float64) - _np(mu)) @ _np(Linv).T).astype(npdt).astype(np.float64)
        return {"x": x, "y": y, "beta": beta, "mu": _np(mu), "L": _np(L), "seed": seed, "gid0": gid0, "target": target_accept,
                "adapt": adapt, "nu": nu, "npdt": npdt, "prop": None, "noise": noise}

    def pcn_ysplit_propose(self, sess, step):
        y, nu, seed, gid0 = sess["y"], sess["nu"], sess["seed"], sess["gid0"]
        rho = self._split["rho"]
        n, d = y.shape
        xi = np.stack([O.pcn_noise(seed, gid0 + i, step, d, sess["noise"])[0] for i in range(n)])
        q0 = (y * y).sum(1)
        rs = np.full(n, rho)
        if nu > 0.0:
            g = np.array([O.gamma_unit(0.5 * (d + nu), seed, gid0 + i, step, sess["noise"]) for i in range(n)])
            rs = rho * np.sqrt((nu + q0) / (2.0 * g))
        yp = (np.sqrt(1 - rho * rho) * y + rs[:, None] * xi).astype(sess["npdt"]).astype(np.float64)
        sess["prop"] = (step, yp, q0, (yp * yp).sum(1))
        xp = (sess["mu"] + yp @ sess["L"].T).astype(sess["npdt"])
        return torch.from_numpy(xp)

    def pcn_ysplit_accept(self, sess, step, ll, lp, lq, ll_new, lp_new, lq_new, n_global, t, logj=None, logj_new=None):
        pstep, yp, q0, q1 = sess["prop"]
        assert pstep == step
        beta, nu, d = sess["beta"], sess["nu"], yp.shape[1]
        n = yp.shape[0]
        u = np.array([O.pcn_noise(sess["seed"], sess["gid0"] + i, step, d, sess["noise"])[1] for i in range(n)])

        def lpt(a, b, c):
            with np.errstate(all="ignore"):
                r = (1 - beta) * _np(c) + beta * (_np(a) + _np(b))
            return np.where(r < np.inf, r, -np.inf)  # NaN and +inf -> -inf (csrc/asmc_pcn_dev.h: log_p_t)

        c0, c1 = (O.tpcn_corr(q0, d, nu), O.tpcn_corr(q1, d, nu)) if nu > 0.0 else (0.5 * q0, 0.5 * q1)
        new, old = lpt(ll_new, lp_new, lq_new), lpt(ll, lp, lq)
        if logj is not None:
            with np.errstate(all="ignore"):
                new, old = new + _np(logj_new), old + _np(logj)
            new, old = np.where(new < np.inf, new, -np.inf), np.where(old < np.inf, old, -np.inf)
        with np.errstate(all="ignore"):
            acc = np.log(u) < (new + c1) - (old + c0)
        sess["y"][acc] = yp[acc]
        acc_t = torch.from_numpy(acc)
        ll[acc_t], lp[acc_t], lq[acc_t] = ll_new[acc_t], lp_new[acc_t], lq_new[acc_t]
        if logj is not None:
            logj[acc_t] = logj_new[acc_t]
        self._split["last"] = int(acc.sum())
        self.pcn_split_adapt(n_global, sess["target"], t, sess["adapt"])

    def pcn_ysplit_end(self, sess, n_steps):
        xs = (sess["mu"] + sess["y"] @ sess["L"].T).astype(sess["npdt"])
        sess["x"].copy_(torch.from_numpy(xs))
        return self.pcn_split_end(n_steps)

    def pcn_accept(self, x, x_prop, ll, lp, lq, ll_new, lp_new, lq_new, q0, q1, beta, seed, gid0, step,
                   logj_old=None, logj_new=None, want_count=True):
        n, d = x.shape
        u = np.array([O.pcn_noise(seed, gid0 + i, step, d)[1] for i in range(n)])

        def lpt(a, b, c):
            with np.errstate(all="ignore"):
                r = (1 - beta) * _np(c) + beta * (_np(a) + _np(b))
            return np.where(r < np.inf, r, -np.inf)  # NaN and +inf -> -inf (csrc/asmc_pcn_dev.h: log_p_t)

        new, old = lpt(ll_new, lp_new, lq_new), lpt(ll, lp, lq)
        with np.errstate(all="ignore"):  # k_pcn_accept_flags: the log-Jacobian joins the log-target, then the guard again
            if logj_new is not None:
                new = new + _np(logj_new)
                new = np.where(new < np.inf, new, -np.inf)
            if logj_old is not None:
                old = old + _np(logj_old)
                old = np.where(old < np.inf, old, -np.inf)
        with np.errstate(all="ignore"):
            acc = np.log(u) < (new + 0.5 * _np(q1)) - (old + 0.5 * _np(q0))
        acc_t = torch.from_numpy(acc)
        x[acc_t] = x_prop[acc_t]
        ll[acc_t], lp[acc_t], lq[acc_t] = ll_new[acc_t], lp_new[acc_t], lq_new[acc_t]
        if logj_old is not None and logj_new is not None:
            logj_old[acc_t] = logj_new[acc_t]
        if not want_count:
            self._split["last"] = int(acc.sum())
            return None
        return int(acc.sum())
