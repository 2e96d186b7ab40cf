#!/usr/bin/env python
"""bench.py — throughput of the SMC particle-batch hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload (default, BASELINE.json configs[1]): 1M particles per GPU, d=32, Gaussian target
(ll = lp = -|x|^2/2), analytic proposal q = N(0, 1.5^2 I) ("flow off"), IS-only: one *step* is one
temperature iteration of the reference loop (smc/base.py:401-445) without mutation:
  adaptive beta bisection (target efficiency 0.5, tol 1e-6) -> ESS(beta), ESS(1) -> evidence ratio +
  variance -> resample (normalised weights, cdf, PCG64 uniforms, search, row gather).
Inputs are resident in HBM before the timed region; every step processes the same pristine batch.
`value` = particles x steps / time ("particle-steps/s"; here a step is a temperature iteration —
SURVEY.md §8d calls this unit particle-iterations/s).

Extra (same JSON line, `extra`): the mutation path — fused pCN kernel throughput and one full
`HipSMC.sample()` run with its log-evidence error for the analytic proposal, and configs[2] proper
(coupling-flow proposal evaluated on the fp32 MFMA inside the device-side pCN loop).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--x-dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--resample-mode", choices=["exact", "fast"], default="exact")
    ap.add_argument("--mcmc-steps", type=int, default=32, help="pCN steps per temperature in the extra leg")
    ap.add_argument("--noise", choices=["f64", "f32"], default="f32", help="proposal-noise generator of the pCN kernel")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-flow-leg", action="store_true", help="skip the coupling-flow (configs[2]) extra leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-extras", action="store_true", help="N > 1: also run the full-sampler extra legs")
    ap.add_argument("--force-sharded", action="store_true",
                    help="test rig: run the sharded code path (process group, collectives, owner layout) with the ranks at hand, "
                         "also when that is a single rank - one GPU then shows the cost of the sharded machinery itself over RCCL")
    ap.add_argument("--shard-layout", choices=["owner", "slots"], default="owner",
                    help="N > 1: offspring stay on the ancestor's rank (default) or single-rank slot order with row exchange")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # test rig only: ASMC_BENCH_BACKEND=gloo ASMC_BENCH_DEVICE=0 runs several ranks on ONE GPU (collectives staged
    # through the host) to exercise the sharded code path where no multi-GPU node is at hand
    backend = os.environ.get("ASMC_BENCH_BACKEND", "nccl")
    if "ASMC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ASMC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from aspire_amd import smc_math
    from aspire_amd.comm import default_comm
    from aspire_amd.engine import HipEngine
    from aspire_amd.flows import GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.samples import gather_global
    from aspire_amd.targets import DiagGaussianMixture

    n_local, d = args.particles_per_gpu, args.dims
    n_global = n_local * world
    xdt = torch.float64 if args.x_dtype == "f64" else torch.float32
    s_bytes = 8 if args.x_dtype == "f64" else 4
    eng = HipEngine(local_rank, n_max=n_global, d_max=max(d, 32))  # the replicated exact cdf scan covers all N
    comm = default_comm(eng.device)
    if sharded and world == 1:  # --force-sharded on one rank: the real communicator over a one-rank group
        from aspire_amd.comm import TorchDistComm

        comm = TorchDistComm(eng.device if backend == "nccl" else torch.device("cpu"))

    # ---- synthetic batch, resident in HBM ---------------------------------------------------
    sigma_q = 1.5
    flow = GaussianFlow(d, sigma=sigma_q, seed=0, engine=eng, dtype=xdt)
    flow.gid0 = rank * n_local
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    x, lq = flow.sample_and_log_prob(n_local)
    ll = eng.mixture_logpdf(x, lik.device_mixture(eng))
    lp = ll.clone()
    torch.cuda.synchronize()

    rng = np.random.default_rng(12345)
    scal = {}

    def is_step():
        m_one = {}

        def eff_fn(betas, closed_form=False):  # same as SMCSampler.determine_beta's (samplers/smc.py)
            if closed_form and "m" in m_one:
                shifts = [m_one["m"] * b for b in betas]
                sts = smc_math.global_stats(eng, comm, ll, lp, lq, 0.0, betas, n_global, shifts=shifts)
            else:
                sts = smc_math.global_stats(eng, comm, ll, lp, lq, 0.0, betas, n_global)
                if len(betas) == 1 and betas[0] == 1.0:
                    m_one["m"] = sts[0].m
            return [smc_math.ess(s) / n_global for s in sts]

        found = {}

        # whole bisection on device: asmc_find_beta, or its sharded form (reduce -> all-gather -> decide per round)
        def search_fn(b0, target, tol):
            if not sharded:
                b, _, conv, passes, n_nan, trip, trip_one = eng.find_beta(ll, lp, lq, b0, target, tol)
            else:
                b, _, conv, passes, n_nan, trip, trip_one = smc_math.find_beta_sharded(eng, comm, ll, lp, lq, b0, target,
                                                                                      tol, n_global)
            assert conv and n_nan == 0
            found.update(beta=b, trip=trip, one=trip_one)
            return b, passes

        beta, _, n_pass = smc_math.determine_beta(eff_fn, 0.0, adaptive=True, beta_step=float("nan"), min_beta_step=0.0,
                                                  max_beta_step=1.0, beta_tolerance=1e-6, adaptive_min_beta_step=False,
                                                  target=0.5, rate=1.0, search_fn=search_fn)
        # same order as the sampler loop (smc/base.py:401-445): ESS(beta), ESS(1), evidence ratio + variance, resample;
        # the device-side search has already reduced the batch at beta* and at 1.0
        if found.get("beta") == beta and found.get("trip") is not None:
            st_b, st_1 = smc_math.Stats(*found["trip"], n_global), smc_math.Stats(*found["one"], n_global)
        else:
            st_b, st_1 = smc_math.global_stats(eng, comm, ll, lp, lq, 0.0, [beta, 1.0], n_global)
        if sharded and args.shard_layout == "owner":
            # offspring stay on the ancestor's rank: one all-gather (rank totals + variance partials), no row exchange
            idx, var, s1p = smc_math.resample_owner(eng, comm, ll, lp, lq, 0.0, beta, n_global, rng,
                                                    mode=args.resample_mode, st=st_b)
            if idx is not None:
                scal.update(beta=beta, ess=smc_math.ess(st_b), ess1=smc_math.ess(st_1),
                            ratio=smc_math.log_evidence_ratio(st_b), var=var, passes=n_pass, layout="owner")
                return eng.gather(idx, x, ll, lp, lq)
        else:
            var, s1p = smc_math.evidence_variance_and_lse(eng, comm, ll, lp, lq, 0.0, beta, st_b)
        scal.update(beta=beta, ess=smc_math.ess(st_b), ess1=smc_math.ess(st_1), ratio=smc_math.log_evidence_ratio(st_b),
                    var=var, passes=n_pass, layout="slots")
        idx, _ = smc_math.resample_indices(eng, comm, ll, lp, lq, 0.0, beta, n_global, rng, mode=args.resample_mode,
                                           st=st_b, s1p=s1p)
        return gather_global(eng, comm, idx, x, ll, lp, lq)

    def sync_all():
        if sharded:
            comm.barrier()
        torch.cuda.synchronize()

    # untimed pre-warm (allocator pools, lazy code-object loads, one-off runtime stalls observed around the
    # 15th-50th iteration of a fresh process: a single ~50 ms hiccup of the runtime), then the W warm-up steps
    for _ in range(150 + args.warmup):
        is_step()
    sync_all()
    prof = None
    if os.environ.get("ASMC_BENCH_CPROFILE") and rank == 0:  # host-side hot spots of the step (stderr)
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = is_step()
    sync_all()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(18)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = n_global * args.steps / dt

    # ---- roofline of the dominant kernel -----------------------------------------------------------
    # The same steps once more with the library's per-kernel HIP events switched on (events recorded on the
    # launch stream around every kernel; kept out of the timed region because two event records per launch
    # perturb a launch-bound loop).  achieved = algorithmic bytes per launch / average kernel duration.
    eng.profile(True)
    n_prof = min(args.steps, 10)
    for _ in range(n_prof):
        is_step()
    kern = eng.profile_report()
    eng.profile(False)
    row_b = d * s_bytes
    alg_bytes = {  # per launch, from SURVEY.md §8d's per-particle figures
        "k_weights_max": 24 * n_local, "k_weights_sums": 24 * n_local, "k_weights_m2": 24 * n_local,
        "k_bis_sums": 24 * n_local, "k_weights_m2_lse": 24 * n_local,
        "k_weights_map": 32 * n_local, "k_tile_sum": 8 * n_local, "k_exact_tile_td_launch": 8 * n_local,
        "k_exact_tile_write": 16 * n_local, "k_tile_scan": 16 * n_local, "k_divide": 16 * n_local,
        "k_pcg64_uniforms": 8 * n_local, "k_pcg64_select": 8 * n_local, "k_search": 24 * n_local, "k_gather16": (2 * (row_b + 24) + 8) * n_local,
        # serial dependency chain over tile records (+ the ~log2 N tiles that straddle a binade): latency bound
        "k_exact_chain": 40 * ((n_local + 2047) // 2048) + 16 * 2048 * 12,
    }

    def alg_of(name):
        base = name.split("<")[0]
        return alg_bytes.get(base)

    tot = {k: c * ms for k, (c, ms) in kern.items() if alg_of(k)}
    dom = max(tot, key=tot.get)
    dom_ms = kern[dom][1]
    achieved = alg_of(dom) / (dom_ms * 1e-3) / 1e9
    traffic = None
    try:  # HBM bytes per launch from the committed PMC run of this configuration (profiles/, separate passes)
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_launch.json")))
        key = f"{dom.split('(')[0]}|n={n_local}|d={d}|{args.x_dtype}"
        traffic = tr.get(key)
    except Exception:
        traffic = None
    per_kernel = {}
    for k, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        ab = alg_of(k)
        per_kernel[k] = {"launches_per_step": round(c / n_prof, 2), "avg_us": round(ms * 1e3, 2),
                         "alg_GBs": None if not ab else round(ab / (ms * 1e-3) / 1e9, 1)}
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_ms": round(dom_ms, 5),
                "alg_bytes_per_launch": alg_of(dom), "gpu_busy_ms_per_step": round(sum(tot.values()) / n_prof, 4),
                "per_kernel": per_kernel}

    result = {
        "metric": "particle-steps/sec (N x n_steps), 1M particles d=32; log-evidence err vs ref",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: IS-only temperature iteration (bisection+ESS+evidence+resample), "
                               f"{n_local} particles/GPU, d={d}, Gaussian target, analytic proposal N(0,1.5^2 I)",
                   "n_steps_meaning": "temperature iterations (no mutation)", "particles_per_gpu": n_local,
                   "global_particles": n_global, "dims": d, "x_dtype": args.x_dtype, "resample_mode": args.resample_mode,
                   "resample_method": "multinomial", "beta_tolerance": 1e-6, "target_efficiency": 0.5,
                   "parallelism": f"particle-shard x{world}" + (f" ({scal.get('layout')} layout)" if sharded else "")},
        "roofline": roofline,
        "scalars": {k: (v if isinstance(v, (int, str)) else float(v)) for k, v in scal.items()},
    }

    # ---- extra: mutation path -------------------------------------------------------------------
    if not args.no_extra:
        extra = {}
        mu0 = eng.asarray(np.zeros(d))
        eye = eng.asarray(np.eye(d))
        xm, llm, lpm, lqm = out[0].clone(), out[1].clone(), out[2].clone(), out[3].clone()
        tgt, qm = lik.device_mixture(eng), flow.device_mixture(eng)
        n_mc = args.mcmc_steps
        eng.pcn_mutate(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, tgt, tgt, qm, 7, rank * n_local, 0.3, 4, 0, 0.234, True, args.noise)
        sync_all()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        n_acc, rho_hist, rho = eng.pcn_mutate(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, tgt, tgt, qm, 7,
                                              rank * n_local, 0.3, n_mc, 4, 0.234, True, args.noise)
        ev1.record()
        sync_all()
        ms = ev0.elapsed_time(ev1) / n_mc
        b_step = (2 * d * s_bytes + 16) * n_local
        # the same loop with the Student-t reference (tpCN, nu = 8): + one gamma-variate kernel per step
        eng.pcn_mutate(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, tgt, tgt, qm, 7, rank * n_local, 0.3, 4, 0, 0.234, True,
                       args.noise, 8.0)
        sync_all()
        ev0.record()
        eng.pcn_mutate(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, tgt, tgt, qm, 7, rank * n_local, 0.3, n_mc, 4, 0.234, True,
                       args.noise, 8.0)
        ev1.record()
        sync_all()
        ms_t = ev0.elapsed_time(ev1) / n_mc
        extra["tpcn_kernel"] = {"ms_per_step": round(ms_t, 4), "particle_steps_per_s_per_gpu": n_local / (ms_t * 1e-3), "nu": 8.0,
                                "noise": args.noise}
        extra["pcn_kernel"] = {"ms_per_step": round(ms, 4), "particle_steps_per_s_per_gpu": n_local / (ms * 1e-3),
                               "alg_bytes_per_step": b_step, "achieved_GBs": round(b_step / (ms * 1e-3) / 1e9, 1),
                               "frac_of_hbm_peak": round(b_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "mean_accept": float(n_acc.mean() / n_local), "rho_final": rho, "noise": args.noise}
        result["extra"] = extra
    # The full-sampler legs issue many more collectives than the headline step; at N > 1 they only run on request, so
    # that the scaling line cannot be lost to them (tools/rig2.sh exercises them with two ranks on one GPU).
    if not args.no_extra and (world == 1 or args.sharded_extras):
        # one full sampler run (configs[2] shape with the analytic proposal): log-evidence check
        sp = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=sigma_q, seed=1, engine=eng, dtype=xdt),
                    xp=np, engine=eng, comm=comm, rng=np.random.default_rng(2), dtype=args.x_dtype.replace("f", "float"))
        sync_all()
        t0 = time.perf_counter()
        post = sp.sample(n_global, sampler_kwargs=dict(n_steps=n_mc, noise=args.noise, step_fn="pcn"), store_sample_history=False,
                         resample_mode=args.resample_mode)
        sync_all()
        t_s = time.perf_counter() - t0
        n_temps = len(sp.history.beta)
        true_logz = 0.5 * d * math.log(math.pi)
        extra["smc_pcn_run"] = {"wall_s": round(t_s, 4), "temperatures": n_temps, "mcmc_steps_per_temperature": n_mc,
                                "particle_steps_per_s": n_global * n_temps * n_mc / t_s,
                                "log_evidence": float(post.log_evidence), "log_evidence_error": float(post.log_evidence_error),
                                "analytic_log_evidence": true_logz,
                                "abs_err_in_sigma": abs(float(post.log_evidence) - true_logz) / max(float(post.log_evidence_error), 1e-300),
                                "mean_accept": float(np.mean(sp.history.mcmc_acceptance))}
        # the reference's default mutation kernel (step_fn="tpcn": Student-t reference refitted per temperature)
        spt = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=GaussianFlow(d, sigma=sigma_q, seed=1, engine=eng, dtype=xdt),
                     xp=np, engine=eng, comm=comm, rng=np.random.default_rng(2), dtype=args.x_dtype.replace("f", "float"))
        spt.sample(min(n_global, 65536 * world), sampler_kwargs=dict(n_steps=2, noise=args.noise), store_sample_history=False,
                   resample_mode=args.resample_mode)  # warm (first-launch costs, scipy import)
        sync_all()
        t0 = time.perf_counter()
        postt = spt.sample(n_global, sampler_kwargs=dict(n_steps=n_mc, noise=args.noise), store_sample_history=False,
                           resample_mode=args.resample_mode)
        sync_all()
        t_t = time.perf_counter() - t0
        extra["smc_tpcn_run"] = {"wall_s": round(t_t, 4), "temperatures": len(spt.history.beta), "mcmc_steps_per_temperature": n_mc,
                                 "particle_steps_per_s": n_global * len(spt.history.beta) * n_mc / t_t,
                                 "log_evidence": float(postt.log_evidence), "log_evidence_error": float(postt.log_evidence_error),
                                 "abs_err_in_sigma": abs(float(postt.log_evidence) - true_logz) / max(float(postt.log_evidence_error), 1e-300),
                                 "nu_per_temperature": [round(v, 2) for v in spt.history.mcmc_nu],
                                 "mean_accept": float(np.mean(spt.history.mcmc_acceptance))}
        # configs[2] proper: coupling-flow proposal (4 coupling layers, MLP 16->64->64->32, float32) — flow
        # log-density on the fp32 MFMA inside the device-side pCN loop; roofline of that kernel against the
        # dense fp32 MFMA peak
        if not args.no_flow_leg:
            from aspire_amd.flows import CouplingFlow

            cflow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
            gtrain = np.random.default_rng(3)
            cflow.fit(sigma_q * 0.9 * gtrain.normal(size=(8000, d)), n_epochs=8)
            sp3 = HipSMC(log_likelihood=lik, log_prior=lik, dims=d, prior_flow=cflow, xp=np, engine=eng, comm=comm,
                         rng=np.random.default_rng(2), dtype=args.x_dtype.replace("f", "float"))
            sp3.sample(min(n_global, 65536 * world), sampler_kwargs=dict(n_steps=2, noise=args.noise, step_fn="pcn"), store_sample_history=False,
                       resample_mode=args.resample_mode)  # warm (first-launch costs)
            sync_all()
            eng.profile(True)
            t0 = time.perf_counter()
            post3 = sp3.sample(n_global, sampler_kwargs=dict(n_steps=n_mc, noise=args.noise, step_fn="pcn"), store_sample_history=False,
                               resample_mode=args.resample_mode)
            sync_all()
            t3 = time.perf_counter() - t0
            rep3 = eng.profile_report()
            eng.profile(False)
            nt3 = len(sp3.history.beta)
            flow_ms = rep3.get("k_coupling_logprob", (0, 0.0))[1]
            flow_flops = n_local * 4 * 2 * ((d // 2) * 64 + 64 * 64 + 64 * d)
            step_ms = sum(rep3.get(k, (0, 0.0))[1] for k in ("k_pcn_flow_propose", "k_coupling_logprob", "k_pcn_flow_accept", "k_pcn_adapt"))
            extra["smc_pcn_flow_run"] = {
                "wall_s": round(t3, 4), "temperatures": nt3, "mcmc_steps_per_temperature": n_mc,
                "particle_steps_per_s": n_global * nt3 * n_mc / t3,
                "log_evidence": float(post3.log_evidence), "log_evidence_error": float(post3.log_evidence_error),
                "analytic_log_evidence": true_logz,
                "abs_err_in_sigma": abs(float(post3.log_evidence) - true_logz) / max(float(post3.log_evidence_error), 1e-300),
                "mean_accept": float(np.mean(sp3.history.mcmc_acceptance)),
                "device_ms_per_mcmc_step": round(step_ms, 4),
                "flow_kernel": {"bound": "mfma", "dtype": "f32", "avg_ms": round(flow_ms, 4), "flops_per_launch": flow_flops,
                                "achieved_TFLOPs": round(flow_flops / (flow_ms * 1e-3) / 1e12, 1) if flow_ms else None,
                                "peak_TFLOPs": 157.3,
                                "frac": round(flow_flops / (flow_ms * 1e-3) / 1e12 / 157.3, 4) if flow_ms else None},
            }
        result["extra"] = extra

    # ---- CPU baseline: the oracle (kind "port") on a bounded sample, rank 0, N=1 only ---------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O

        n_cpu = min(n_local, 1_000_000)
        xc = x[:n_cpu].double().cpu().numpy()
        llc, lpc, lqc = (t[:n_cpu].cpu().numpy() for t in (ll, lp, lq))
        st = O.pcg64_state_from_numpy(np.random.default_rng(12345))
        O.is_iteration(xc[:1000], llc[:1000], lpc[:1000], lqc[:1000], 0.0, 0.5, 1e-6, st.copy())  # warm the library
        t0 = time.perf_counter()
        n_it = 0
        while True:
            (_, _, _, _), sc = O.is_iteration(xc, llc, lpc, lqc, 0.0, 0.5, 1e-6, st)
            n_it += 1
            el = time.perf_counter() - t0
            if el > 12.0 or n_it >= 20:
                break
        result["cpu_baseline"] = {"value": n_cpu * n_it / el, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                                  "sample": f"{n_it} IS-only temperature iterations of the same {n_cpu} x {d} fp64 batch "
                                            f"(oracle/asmc_oracle.c orc_is_iteration, single thread)",
                                  "host_cpus": os.cpu_count(), "beta": float(sc[0])}
        result["cpu_baseline"]["beta_matches_gpu"] = bool(sc[0] == scal["beta"])
    if rank == 0:
        print(json.dumps(result))
    if sharded:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
