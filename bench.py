#!/usr/bin/env python
"""bench.py — throughput of the SMC particle-batch hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched under
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1M particles per GPU,
d = 32, Gaussian target (ll = lp = -|x|^2/2), RealNVP coupling-flow proposal (4 coupling layers, MLP 16->64->64->32,
float32, trained once before the timed region) whose log-density runs on the fp32 MFMA inside the mutation loop,
adaptive tempering (target efficiency 0.5, tolerance 1e-6), exact multinomial resampling, pCN mutation with 32 steps
per temperature and the sampler's DEFAULT fp64 proposal noise.
One bench *step* = one complete `HipSMC.sample()` run (proposal draw, ~6 temperatures x [beta search, ESS, evidence,
resample, reference fit, 32 mutation steps]).  `value` = particles x mutation steps executed / wall time of the K
runs = particle-steps/s as SURVEY.md §8d defines it (N x number of mutation steps / wall of sample()).
Particles are sharded over the ranks (weak scaling: 1M per GPU); all inputs are generated on the device.

Same JSON line: `roofline` (the flow kernel against the fp32 MFMA peak; the pCN kernels against the HBM peak; the
whole mutation step against both), `cpu_baseline` (the C oracle's restatement of the same mutation step on the host's
cores, OpenMP), `extra` (IS-only temperature iteration of configs[1], the analytic-proposal runs, f32-noise variants).
"""
from __future__ import annotations

import argparse
import gc
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
MFMA_F32_PEAK_TF = 157.3  # dense fp32-input MFMA peak (MI355X_MICROARCH.md, matrix cores)
MFMA_F16_PEAK_TF = 2516.6  # dense fp16 MFMA peak (same table)


def kernel_source_hash() -> str:
    """sha256 (16 hex digits) over the sources of the dominant kernel: the fused flow-proposal step, the flow layers and the
    proposal noise.  profiles/traffic_per_launch.json and profiles/sq_counters.json are stamped with it."""
    import hashlib

    h = hashlib.sha256()
    for name in ("asmc_pcn_fused.hip", "asmc_flow_dev.h", "asmc_pcn_dev.h"):
        with open(os.path.join(ROOT, "aspire_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def launch_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: this process becomes the launcher.  It starts N fresh
    copies of itself, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as
    `torch.distributed.run` would set them), relays rank 0's stdout - the ONE JSON line - and returns the worst exit status.
    It never touches the GPU itself (no HIP call, no `torch.cuda.is_available()`: only `device_count()`, which does not
    initialise the runtime on this image) and never re-execs.  When the ranks cannot be started (fewer visible GPUs than ranks)
    it returns non-zero instead of measuring one GPU: a line with `n_gpus: 1` under `--gpus 8` would void a scaling run.
    Test rig: ASMC_BENCH_DEVICE=<i> (with ASMC_BENCH_BACKEND=gloo) puts every rank on GPU i."""
    import socket
    import subprocess

    if "ASMC_BENCH_DEVICE" not in os.environ:
        have = torch.cuda.device_count()
        if have < n_gpus:
            print(f"bench.py: --gpus {n_gpus} but {have} HIP device(s) visible; refusing to measure fewer GPUs than asked for",
                  file=sys.stderr)
            return 2
    import threading

    for attempt in range(3):
        with socket.socket() as sk:  # a free rendezvous port
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(n_gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd(),
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=subprocess.PIPE if r == 0 else None,
                                          text=True if r == 0 else None))
        got, err0 = [], []
        reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
        ereader = threading.Thread(target=lambda: err0.append(procs[0].stderr.read()), daemon=True)  # rank 0 hosts the rendezvous
        reader.start(), ereader.start()
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            failed = next((p for p in procs if p.poll() not in (None, 0)), None)
            time.sleep(0.05)
        if failed is not None:  # one rank died: its peers would wait in a collective until a watchdog fires - end exactly those processes
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        rcs = [p.wait() for p in procs]
        reader.join(timeout=10), ereader.join(timeout=10)
        out, err = (got[0] if got else ""), (err0[0] if err0 else "")
        # the port was free when it was picked, not necessarily when rank 0 listened on it: that one failure is repeated on a fresh port
        if any(rcs) and "EADDRINUSE" in err and attempt < 2:
            print("bench.py: rendezvous port taken before rank 0 could listen on it; starting the ranks again", file=sys.stderr)
            continue
        sys.stderr.write(err)
        break
    worst = next((rc if rc > 0 else 1 for rc in rcs if rc != 0), 0)
    lines = [ln for ln in (out or "").splitlines() if ln.strip()]
    if worst == 0 and len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} lines, expected one", file=sys.stderr)
        worst = 3
    if worst == 0:
        sys.stdout.write(lines[0] + "\n")
        sys.stdout.flush()
    else:
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
    return worst


def derive_limiter(sq: dict | None, hbm_frac: float | None, mfma_frac_executed: float | None) -> dict:
    """What bounds the dominant kernel, derived from the numbers the line carries: the SQ counters of the committed profile (fractions
    of SIMD time / of wave cycles; None when the kernel sources have moved on since they were collected) and the two measured
    rates of this run.  The largest share names the limiter; the shares and the rule travel with it."""
    shares = {}
    if sq:
        for key, label in (("valu_active", "vector-ALU issue"), ("mfma_busy", "matrix pipe busy"),
                           ("wait_any_of_wave_cycles", "waves parked at s_waitcnt / barriers"),
                           ("wait_inst_any_of_wave_cycles", "issue stalls (dependencies, pipe busy)")):
            if sq.get(key) is not None:
                shares[label] = float(sq[key])
    if hbm_frac is not None:
        shares["HBM bandwidth (algorithmic bytes / peak)"] = float(hbm_frac)
    if mfma_frac_executed is not None:
        shares["matrix-pipe arithmetic (executed flops / dense peak)"] = float(mfma_frac_executed)
    if not shares:
        return {"name": None, "shares": {}, "rule": "no counters"}
    name = max(shares, key=shares.get)
    v, m = shares.get("vector-ALU issue"), shares.get("matrix pipe busy")
    out = {"name": name, "shares": {k: round(x, 4) for k, x in sorted(shares.items(), key=lambda kv: -kv[1])},
           "rule": "largest share; SQ shares are fractions of SIMD time (valu_active, mfma_busy) or of wave cycles (waits) from "
                   "profiles/sq_counters.json, the two rates are measured in this run",
           "counters_current": bool(sq)}
    if v is not None and m is not None:
        # the mix bound of DESIGN 3.1: an fp16 MFMA hides about half of a partner wave's vector work, so time >= V + other issue + M / 2
        out["valu_plus_half_mfma"] = round(v + 0.5 * m, 4)
    return out


def main():
    # `--gpus N` without a launcher around it: become the launcher, before anything touches the GPU
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        n_req = pre.parse_known_args()[0].gpus
        if n_req > 1:
            raise SystemExit(launch_ranks(n_req))
    # stdout carries ONE line - the JSON record.  Libraries write banners there (RCCL prints its version block when a communicator
    # is made): from here on file descriptor 1 is stderr, and the record goes to the descriptor stdout had.
    sys.stdout.flush()
    record_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed HipSMC.sample() runs")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--x-dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--resample-mode", choices=["exact", "fast"], default="exact")
    ap.add_argument("--mcmc-steps", type=int, default=32, help="pCN steps per temperature")
    ap.add_argument("--noise", choices=["f64", "f32"], default="f64", help="proposal-noise generator (sampler default: f64)")
    ap.add_argument("--step-fn", choices=["pcn", "tpcn"], default="pcn")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded-extras", action="store_true", help="N > 1: also run the extra legs")
    ap.add_argument("--force-sharded", action="store_true",
                    help="test rig: run the sharded code path (process group, RCCL collectives, owner layout) with a ONE-rank "
                         "group - one GPU then shows the cost of the sharded machinery itself")
    ap.add_argument("--shard-layout", choices=["owner", "slots"], default="owner",
                    help="N > 1: offspring stay on the ancestor's rank (default) or single-rank slot order with row exchange")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Host-side pauses of the interpreter's cycle collector are recorded (generation, ms) so that a stall inside a timed leg
    # can be attributed: round 3's driver run of the IS-only leg lost ~80 ms of its 91 ms to ONE such event.
    gc_log, gc_t0 = [], [0.0]

    def _gc_cb(phase, info):
        if phase == "start":
            gc_t0[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], (time.perf_counter() - gc_t0[0]) * 1e3))

    gc.callbacks.append(_gc_cb)

    def gc_summary(since: int):
        ev = gc_log[since:]
        full = [ms for g, ms in ev if g == 2]
        return {"collections": len(ev), "full_collections": len(full), "total_ms": round(sum(ms for _, ms in ev), 3),
                "max_pause_ms": round(max((ms for _, ms in ev), default=0.0), 3)}
    if world != args.gpus:  # (a launcher's WORLD_SIZE that disagrees with --gpus: never measure another job than the one asked for)
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
    from aspire_amd.flows import CouplingFlow, GaussianFlow
    from aspire_amd.samplers.smc import HipSMC
    from aspire_amd.targets import DiagGaussianMixture

    n_local, d = args.particles_per_gpu, args.dims
    n_global = n_local * world
    xdt = torch.float64 if args.x_dtype == "f64" else torch.float32
    xname = args.x_dtype.replace("f", "float")
    s_bytes = 8 if args.x_dtype == "f64" else 4
    n_mc = args.mcmc_steps
    eng = HipEngine(local_rank, n_max=n_global, d_max=max(d, 32))  # sharded: the draws of all ranks are walked on every rank
    comm = default_comm(eng.device)
    if sharded and world == 1:  # --force-sharded: the real communicator over a one-rank group
        from aspire_amd.comm import TorchDistComm

        comm = TorchDistComm(eng.device if backend == "nccl" else torch.device("cpu"))
        comm.force_sharded = True

    # ---- the workload: targets, trained proposal flow -------------------------------------------------------------
    sigma_q = 1.5
    lik = DiagGaussianMixture.isotropic(d, normalized=False)
    true_logz = 0.5 * d * math.log(math.pi)
    cflow = CouplingFlow(d, n_layers=4, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
    cflow.fit(sigma_q * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)  # untimed

    def sync_all():
        if sharded:
            comm.barrier()
        torch.cuda.synchronize()

    flow_math = "f32-mfma" if os.environ.get("ASMC_FLOW_MATH") == "f32" else "f16x2-split"

    def run(seed: int, n=n_global, flow=cflow, step_fn=args.step_fn, noise=args.noise, steps=n_mc, comm_=None, x_dtype=xname,
            targets=None, xp_=np):
        t_ll, t_lp = targets or (lik, lik)
        sp = HipSMC(log_likelihood=t_ll, log_prior=t_lp, dims=d, prior_flow=flow, xp=xp_, engine=eng, comm=comm_ or comm,
                    rng=np.random.default_rng(seed), dtype=x_dtype)
        sp.shard_layout = args.shard_layout
        post = sp.sample(n, sampler_kwargs=dict(n_steps=steps, noise=noise, step_fn=step_fn), store_sample_history=False,
                         resample_mode=args.resample_mode)
        return sp, post

    run(1, n=min(n_global, 65536 * world), steps=2)  # first-launch costs (code objects, allocator pools, scipy import)
    for w in range(args.warmup):
        run(100 + w)
    # the interpreter's long-lived objects (torch, numpy, scipy, the trained flow: ~1e6 of them) go to the permanent
    # generation: a full collection inside a timed leg then walks only what the legs themselves created (it cost 60-80 ms
    # before; ASMC_BENCH_NO_GC_FREEZE=1 restores that)
    if not os.environ.get("ASMC_BENCH_NO_GC_FREEZE"):
        gc.collect()
        gc.freeze()
    sync_all()
    gc_mark = len(gc_log)
    t0 = time.perf_counter()
    steps_done, zs, temps, accs = 0, [], [], []
    for k in range(args.steps):
        sp, post = run(1000 + k)
        nt = len(sp.history.beta)
        steps_done += nt * n_mc
        temps.append(nt)
        zs.append((float(post.log_evidence) - true_logz) / max(float(post.log_evidence_error), 1e-300))
        accs.append(float(np.mean(sp.history.mcmc_acceptance)))
    sync_all()
    dt = time.perf_counter() - t0
    gc_headline = gc_summary(gc_mark)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = n_global * steps_done / dt

    # ---- roofline: one more run with the library's per-kernel HIP events switched on ---------------------------------
    # (events are recorded on the launch stream around every kernel of the library; kept out of the timed region
    # because two event records per launch perturb a launch-bound loop)
    eng.profile(True)
    sp_p, _ = run(2000)
    kern = eng.profile_report()
    eng.profile(False)
    nt_p = len(sp_p.history.beta)
    n_mut = nt_p * n_mc
    row_b = d * s_bytes
    flow_flops = n_local * 4 * 2 * ((d // 2) * 64 + 64 * 64 + 64 * d)  # 4 coupling layers, MLP d/2 -> 64 -> 64 -> d
    step_kernels = [k for k in kern if k.split("<")[0] in ("k_pcn_flow_propose", "k_coupling_logprob", "k_pcn_flow_accept",
                                                              "k_pcn_adapt", "k_pcn_flow_fused", "k_gamma_draw")]
    step_ms = sum(kern[k][0] * kern[k][1] for k in step_kernels) / max(n_mut, 1)
    flow_k = next((k for k in kern if k.startswith("k_coupling_logprob") or k.startswith("k_pcn_flow_fused")), None)
    flow_ms = kern[flow_k][1] if flow_k else None
    # algorithmic bytes per particle per step: 2 d s + 16 for the pCN state update, + d s + 16 when the
    # proposal density is a flow evaluated by its own pass over the proposed rows (x' written, re-read; log q written, re-read)
    b_pcn = (2 * row_b + 16) * n_local
    b_step = (3 * row_b + 32) * n_local
    # Counters that bench.py cannot collect itself (rocprofv3 --pmc needs its own passes) come from the committed profile
    # files.  They describe ONE build of the kernel: both files carry the hash of the kernel's sources at collection time
    # (tools/stamp_profiles.py) and the fields are null when the tree's sources have moved on.
    src_hash = kernel_source_hash()
    traffic, traffic_src = None, None
    try:  # HBM bytes per launch from the committed PMC run of this configuration (profiles/, separate --pmc passes)
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_per_launch.json")))
        if tr.get("_kernel_source_hash") == src_hash:
            traffic = tr.get(f"{(flow_k or '').split('<')[0]}|n={n_local}|d={d}|{args.x_dtype}")
            traffic_src = tr.get("_comment")
        else:
            traffic_src = "stale: profiles/traffic_per_launch.json was collected on other kernel sources"
    except Exception:
        traffic = None
    per_kernel = {}
    for k, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:24]:
        per_kernel[k] = {"launches_per_run": c, "avg_us": round(ms * 1e3, 2), "share_of_gpu_time": 0.0}
    tot_ms = sum(c * ms for c, ms in kern.values())
    for k in per_kernel:
        per_kernel[k]["share_of_gpu_time"] = round(kern[k][0] * kern[k][1] / tot_ms, 4)
    mfma_floor = flow_flops / (MFMA_F32_PEAK_TF * 1e12) * 1e3
    hbm_floor = b_pcn / (HBM_PEAK_GBS * 1e9) * 1e3
    split = flow_math == "f16x2-split"
    exec_flops = 3 * flow_flops if split else flow_flops          # MFMA flops actually issued per launch
    exec_peak = MFMA_F16_PEAK_TF if split else MFMA_F32_PEAK_TF   # dense peak of the instruction that issues them
    sq, sq_src = None, None
    try:  # SQ counters of the dominant kernel from the committed rocprofv3 --pmc run (profiles/sq_counters.json)
        sqj = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))
        if sqj.get("_kernel_source_hash") == src_hash:
            sq = sqj.get(f"{(flow_k or '').split('<')[0]}|n={n_local}|d={d}|{args.x_dtype}|{args.noise}")
            sq_src = sq.get("source") if sq else None
        else:
            sq_src = "stale: profiles/sq_counters.json was collected on other kernel sources"
    except Exception:
        sq = None
    # accuracy of the flow arithmetic of THIS run (north star: log-weights within 1e-6 relative): the device kernel's log q on a
    # 64k subsample of the run's own final population against the same parameters evaluated in fp64
    flow_rel = None
    try:
        xs_acc = torch.as_tensor(post.x, device=eng.device)
        rows = torch.arange(0, xs_acc.shape[0], max(1, xs_acc.shape[0] // 65536), device=eng.device)[:65536]
        xa = xs_acc[rows].to(torch.float64).contiguous()
        got = eng.coupling_logprob(xa, cflow.device_coupling(eng))
        ref64 = cflow.log_prob_f64(xa)
        rel = (got - ref64).abs() / ref64.abs().clamp_min(1.0)
        flow_rel = {"max_rel": float(rel.max()), "rows": int(rows.numel()), "nonfinite": int((~torch.isfinite(got)).sum()),
                    "what": "asmc_coupling_logprob (the fused step's layer arithmetic) on a strided subsample of the last timed run's "
                            "posterior, against CouplingFlow.log_prob_f64 (same fp32 parameters widened, fp64 arithmetic); "
                            "denominator max(|log q|, 1)"}
    except Exception as exc:  # never lose the line over the side measurement
        flow_rel = {"error": repr(exc)}
    roofline = {
        # the dominant kernel, priced against the pipe it USES ("achieved / peak at the dtype used"): the fp32
        # layers run as split-fp16 products (each fp32 operand an fp16 (hi, lo) pair, three v_mfma_f32_32x32x16_f16 per K = 16
        # with fp32 accumulation, csrc/asmc_flow_dev.h).  THREE fractions, because they answer different questions:
        #   frac_algorithmic  SURVEY §8d's own definition: the flow's 57 kflop per particle / time / dense fp16 peak
        #   frac (executed)   the fp16 MFMA flops actually issued (3 x algorithmic) / time / the same peak
        #   hbm_frac          SURVEY §8d's fused bytes 2 d s + 16 per particle / time / HBM peak
        "bound": "mfma", "kernel": flow_k, "dtype": "f16 (split products of f32 operands)" if split else "f32",
        "flow_math": flow_math,
        "achieved": round(exec_flops / (flow_ms * 1e-3) / 1e12, 2) if flow_ms else None, "peak": exec_peak, "unit": "TFLOP/s",
        "frac": round(exec_flops / (flow_ms * 1e-3) / 1e12 / exec_peak, 4) if flow_ms else None,
        "frac_algorithmic": round(flow_flops / (flow_ms * 1e-3) / 1e12 / exec_peak, 4) if flow_ms else None,
        "achieved_algorithmic": round(flow_flops / (flow_ms * 1e-3) / 1e12, 2) if flow_ms else None,
        "flops_per_launch": exec_flops, "algorithmic_f32_flops_per_launch": flow_flops,
        "hbm_frac": round(b_pcn / (flow_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if flow_ms else None,
        "hbm_achieved_GBs": round(b_pcn / (flow_ms * 1e-3) / 1e9, 1) if flow_ms else None,
        "algorithmic_bytes_per_launch": b_pcn,
        "traffic": traffic, "traffic_source": traffic_src,
        "traffic_over_algorithmic": round(traffic / b_pcn, 3) if traffic else None,
        # what limits it (rocprofv3 --pmc SQ counters of the committed profile, fractions of SIMD time): the vector ALU
        "limiter": derive_limiter(sq, b_pcn / (flow_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if flow_ms else None,
                                  exec_flops / (flow_ms * 1e-3) / 1e12 / exec_peak if flow_ms else None),
        "valu_active": sq.get("valu_active") if sq else None, "mfma_busy": sq.get("mfma_busy") if sq else None,
        "valu_insts_per_64_particle_tile": sq.get("valu_insts_per_tile") if sq else None,
        "sq_counters_source": sq_src, "kernel_source_hash": src_hash,
        "flow_max_rel_vs_fp64": flow_rel,
        "avg_ms": round(flow_ms, 5) if flow_ms else None,
        "whole_step": {"device_ms_per_mutation_step": round(step_ms, 5), "executed_mfma_floor_ms": round(exec_flops / (exec_peak * 1e12) * 1e3, 5),
                       "f32_mfma_floor_ms": round(mfma_floor, 5), "hbm_floor_ms": round(hbm_floor, 5)},
        "gpu_busy_ms_per_run": round(tot_ms, 3), "wall_ms_per_run": round(dt / args.steps * 1e3, 3),
        "gpu_busy_over_wall": round(tot_ms / (dt / args.steps * 1e3), 4),
        "mutation_share_of_gpu_time": round(step_ms * n_mut / tot_ms, 4) if tot_ms else None,
        "per_kernel": per_kernel,
    }

    result = {
        "metric": "particle-steps/sec (N x n_steps), 1M particles d=32; log-evidence err vs ref",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[2]: HipSMC.sample(), {n_local} particles/GPU, d={d}, Gaussian target, RealNVP "
                               f"coupling-flow proposal (4 layers, MLP {d // 2}->64->64->{d}, f32 on the MFMA), adaptive "
                               f"tempering, {n_mc} {args.step_fn} steps per temperature, {args.noise} proposal noise",
                   "step_meaning": "one bench step = one full sample() run; value counts its mutation steps "
                                   "(N x temperatures x mcmc steps / wall)",
                   "particles_per_gpu": n_local, "global_particles": n_global, "dims": d, "x_dtype": args.x_dtype,
                   "flow_dtype": "f32", "flow_math": flow_math, "mcmc_steps_per_temperature": n_mc, "step_fn": args.step_fn, "noise": args.noise,
                   "resample_mode": args.resample_mode, "resample_method": "multinomial", "beta_tolerance": 1e-6,
                   "target_efficiency": 0.5,
                   "parallelism": f"particle-shard x{world}" + (f" ({args.shard_layout} layout)" if sharded else "")
                   + (" [--force-sharded rig: sharded code path over a one-rank RCCL group]" if sharded and world == 1 else "")},
        "mutation_steps_per_run": steps_done / args.steps, "temperatures_per_run": float(np.mean(temps)),
        "ms_per_mutation_step": dt / max(steps_done, 1) * 1e3,
        "log_evidence": {"analytic": true_logz, "z_scores": [round(z, 3) for z in zs],
                         "rms_z": float(np.sqrt(np.mean(np.square(zs)))), "max_abs_z": float(np.max(np.abs(zs))),
                         "within_1_sigma_fraction": float(np.mean(np.abs(zs) <= 1.0))},
        "mean_accept": float(np.mean(accs)),
        "host_gc_in_timed_region": gc_headline,
        "roofline": roofline,
    }

    # ---- extra legs ---------------------------------------------------------------------------------------------
    if not args.no_extra and (world == 1 or args.sharded_extras):
        extra = {}
        # (a) configs[1]: the IS-only temperature iteration (bisection + ESS + evidence + exact resample + gather) on the
        #     pristine analytic-proposal batch, as round 1 timed it
        gflow = GaussianFlow(d, sigma=sigma_q, seed=0, engine=eng, dtype=xdt)
        gflow.gid0 = rank * n_local
        x, lq = gflow.sample_and_log_prob(n_local)
        ll = eng.mixture_logpdf(x, lik.device_mixture(eng))
        lp = ll.clone()
        rng_is = np.random.default_rng(12345)
        scal = {}

        # every device buffer of the leg is allocated ONCE (index vector, two sets of gather destinations used in turn): the
        # timed loop performs no allocation, so allocator state left behind by the headline runs cannot enter the figure
        idx_buf = torch.empty(n_global, dtype=torch.int64, device=eng.device)
        out_bufs = [(torch.empty_like(x), torch.empty_like(ll), torch.empty_like(lp), torch.empty_like(lq)) for _ in range(2)]
        host_t = {"enqueue_search_resample": 0.0, "enqueue_gather": 0.0, "wait_result": 0.0, "python_scalars": 0.0}
        is_calls = [0]

        def is_step():
            k = is_calls[0] = is_calls[0] + 1
            dst = out_bufs[k & 1]
            if not sharded and args.resample_mode == "exact" and hasattr(eng, "importance_step"):
                # the sampler loop's path (SMCSamples.speculate_importance_step): one chain of launches, one sync
                h0 = time.perf_counter()
                idx = eng.importance_step(ll, lp, lq, 0.0, 0.5, 1e-6, smc_math.pcg64_state(rng_is), n_global, idx_out=idx_buf)
                h1 = time.perf_counter()
                rows = eng.gather(idx, x, ll, lp, lq, out=dst)
                h2 = time.perf_counter()
                b, _, conv, _, n_nan, trip, trip_one, m2, _, found = eng.importance_result()
                h3 = time.perf_counter()
                assert conv and found and n_nan == 0
                rng_is.bit_generator.advance(n_global)
                st_b, st_1 = smc_math.Stats(*trip, n_global), smc_math.Stats(*trip_one, n_global)
                mean_u = st_b.S1 / n_global
                scal.update(beta=b, ess=smc_math.ess(st_b), ess1=smc_math.ess(st_1), ratio=smc_math.log_evidence_ratio(st_b),
                            var=(m2 / n_global) / (n_global * mean_u**2))
                h4 = time.perf_counter()
                host_t["enqueue_search_resample"] += h1 - h0
                host_t["enqueue_gather"] += h2 - h1
                host_t["wait_result"] += h3 - h2
                host_t["python_scalars"] += h4 - h3
                return rows
            if sharded:
                b, _, conv, _, n_nan, trip, trip_one = smc_math.find_beta_sharded(eng, comm, ll, lp, lq, 0.0, 0.5, 1e-6, n_global)
            else:
                b, _, conv, _, n_nan, trip, trip_one = eng.find_beta(ll, lp, lq, 0.0, 0.5, 1e-6)
            assert conv and n_nan == 0
            st_b, st_1 = smc_math.Stats(*trip, n_global), smc_math.Stats(*trip_one, n_global)
            if sharded:
                idx, var, _, _ = smc_math.resample_owner(eng, comm, ll, lp, lq, 0.0, b, n_global, rng_is,
                                                         mode=args.resample_mode, st=st_b)
            else:
                var, s1p = smc_math.evidence_variance_and_lse(eng, comm, ll, lp, lq, 0.0, b, st_b)
                idx, _ = smc_math.resample_indices(eng, comm, ll, lp, lq, 0.0, b, n_global, rng_is, mode=args.resample_mode,
                                                   st=st_b, s1p=s1p)
            scal.update(beta=b, ess=smc_math.ess(st_b), ess1=smc_math.ess(st_1), ratio=smc_math.log_evidence_ratio(st_b), var=var)
            return eng.gather(idx, x, ll, lp, lq, out=dst if idx.numel() == n_local else None)

        for _ in range(60):
            is_step()
        sync_all()
        # timed in batches: the figure is the MEDIAN batch (min / max beside it), each batch bracketed by a device sync
        n_batches, n_is = 7, 20
        batch_ms = []
        for key in host_t:
            host_t[key] = 0.0
        gc_mark_is = len(gc_log)
        for _ in range(n_batches):
            sync_all()
            t0 = time.perf_counter()
            for _ in range(n_is):
                out = is_step()
            sync_all()
            batch_ms.append((time.perf_counter() - t0) / n_is * 1e3)
        host_ms = {k: round(v / (n_batches * n_is) * 1e3, 4) for k, v in host_t.items()}
        gc_is = gc_summary(gc_mark_is)
        t_is = float(np.median(batch_ms)) * 1e-3
        eng.profile(True)
        for _ in range(10):
            is_step()
        kis = eng.profile_report()
        eng.profile(False)
        gk = next((k for k in kis if k.startswith("k_gather")), None)
        g_bytes = (2 * (row_b + 24) + 8) * n_local
        busy_is = sum(c * ms for c, ms in kis.values()) / 10
        diag = None
        if t_is * 1e3 > 1.3 * busy_is:
            worst = max(host_ms, key=host_ms.get)
            diag = (f"median wall {t_is * 1e3:.3f} ms > 1.3 x GPU-busy {busy_is:.3f} ms: host side dominates, largest host "
                    f"segment '{worst}' = {host_ms[worst]:.3f} ms per step (wait_result includes the GPU time of the step)")
        extra["is_only_step"] = {
            "workload": "configs[1]: one temperature iteration without mutation on the pristine 1M x 32 batch",
            "ms_per_step": round(t_is * 1e3, 4), "particle_iterations_per_s": n_global / t_is,
            "ms_per_step_batches": {"min": round(min(batch_ms), 4), "median": round(t_is * 1e3, 4), "max": round(max(batch_ms), 4),
                                    "n_batches": n_batches, "steps_per_batch": n_is},
            "host_ms_per_step": host_ms, "host_gc": gc_is,
            "wall_over_gpu_busy": round(t_is * 1e3 / busy_is, 3) if busy_is else None,
            "diagnosis": diag,
            "alg_bytes_per_step": (64 + 2 * (row_b + 24)) * n_local,
            "whole_step_frac_of_hbm_peak": round((64 + 2 * (row_b + 24)) * n_local / t_is / 1e9 / HBM_PEAK_GBS, 4),
            "launches_per_step": round(sum(c for c, _ in kis.values()) / 10, 1),
            "gpu_busy_ms_per_step": round(busy_is, 4),
            "gather": None if gk is None else {"kernel": gk, "avg_us": round(kis[gk][1] * 1e3, 2),
                                               "achieved_GBs": round(g_bytes / (kis[gk][1] * 1e-3) / 1e9, 1),
                                               "frac_of_hbm_peak": round(g_bytes / (kis[gk][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "per_kernel_us": {k: [round(c / 10, 1), round(ms * 1e3, 2)] for k, (c, ms) in
                              sorted(kis.items(), key=lambda kv: -kv[1][0] * kv[1][1])},
            "scalars": {k: float(v) for k, v in scal.items()},
        }
        # (b) the fused pCN kernel on the resampled population: default f64 noise, and the fast f32 generator
        mu0, eye = eng.asarray(np.zeros(d)), eng.asarray(np.eye(d))
        tgt, qm = lik.device_mixture(eng), gflow.device_mixture(eng)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b_step_g = (2 * row_b + 16) * n_local
        for noise in ("f64", "f32"):
            for nu, label in ((0.0, "pcn"), (8.0, "tpcn")):
                xm, llm, lpm, lqm = (t.clone() for t in out)
                eng.pcn_mutate(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, tgt, tgt, qm, 7, rank * n_local, 0.3, 4, 0, 0.234, True, noise, nu)
                sync_all()
                ev0.record()
                n_acc, _, rho = eng.pcn_mutate(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, tgt, tgt, qm, 7, rank * n_local,
                                               0.3, n_mc, 4, 0.234, True, noise, nu)
                ev1.record()
                sync_all()
                ms = ev0.elapsed_time(ev1) / n_mc
                extra[f"{label}_kernel_{noise}_noise"] = {
                    "ms_per_step": round(ms, 4), "particle_steps_per_s_per_gpu": n_local / (ms * 1e-3),
                    "alg_bytes_per_step": b_step_g, "achieved_GBs": round(b_step_g / (ms * 1e-3) / 1e9, 1),
                    "frac_of_hbm_peak": round(b_step_g / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "mean_accept": float(n_acc.mean() / n_local)}
        # (b2) the one-kernel flow-proposal step AWAY from the headline's 98 % acceptance and single-Gaussian targets (round 3's
        #      kernel parked / copied state and slowed down by 28 % at ordinary acceptance rates, and sent mixture targets to
        #      three kernels per step): kernel time from the library's HIP events, 16 steps each on the resampled 1M x 32 batch
        try:
            devc = cflow.device_coupling(eng)
            rob = {}
            mix2 = DiagGaussianMixture(np.stack([0.4 * np.ones(d), -0.4 * np.ones(d)]), np.stack([np.ones(d), 0.7 * np.ones(d)])).device_mixture(eng)
            for label, rho_r, adapt_r, t_lik in (("accept_98pct", 0.02, False, tgt), ("accept_adapted_to_23pct_target", 0.3, True, tgt),
                                                  ("two_component_mixture_likelihood", 0.3, True, mix2)):
                xm, llm, lpm, lqm = (t.clone() for t in out)
                llm = eng.mixture_logpdf(xm, t_lik)
                lqm = eng.coupling_logprob(xm, devc)
                eng.pcn_mutate_flow(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, t_lik, tgt, devc, 7, rank * n_local, rho_r, 4, 0, 0.234, adapt_r, "f64", 0.0)
                eng.profile(True)
                n_acc_r, _, _ = eng.pcn_mutate_flow(xm, llm, lpm, lqm, scal["beta"], mu0, eye, eye, t_lik, tgt, devc, 7, rank * n_local, rho_r, 16, 4,
                                                    0.234, adapt_r, "f64", 0.0)
                kr = eng.profile_report()
                eng.profile(False)
                fk = next((k for k in kr if k.startswith("k_pcn_flow_fused")), None)
                rob[label] = {"kernel": fk, "launches": kr[fk][0] if fk else None, "avg_us": round(kr[fk][1] * 1e3, 2) if fk else None,
                              "mean_accept": float(np.mean(n_acc_r) / n_local),
                              "other_step_kernels": sorted(k for k in kr if k.startswith(("k_coupling_logprob", "k_pcn_flow_propose", "k_pcn_flow_accept")))}
            extra["fused_step_by_regime"] = rob
        except Exception as exc:
            extra["fused_step_by_regime"] = {"error": repr(exc)}
        # (c) full runs with the analytic proposal (no flow): pCN and the reference's default tpCN, default noise
        for label, step_fn in (("smc_pcn_run", "pcn"), ("smc_tpcn_run", "tpcn")):
            run(1, n=min(n_global, 65536 * world), flow=GaussianFlow(d, sigma=sigma_q, seed=1, engine=eng, dtype=xdt),
                step_fn=step_fn, steps=2)
            sync_all()
            t0 = time.perf_counter()
            spx, postx = run(2, flow=GaussianFlow(d, sigma=sigma_q, seed=1, engine=eng, dtype=xdt), step_fn=step_fn)
            sync_all()
            tx = time.perf_counter() - t0
            ntx = len(spx.history.beta)
            extra[label] = {"wall_s": round(tx, 4), "temperatures": ntx, "particle_steps_per_s": n_global * ntx * n_mc / tx,
                            "log_evidence": float(postx.log_evidence), "log_evidence_error": float(postx.log_evidence_error),
                            "abs_err_in_sigma": abs(float(postx.log_evidence) - true_logz) / max(float(postx.log_evidence_error), 1e-300),
                            "mean_accept": float(np.mean(spx.history.mcmc_acceptance)), "noise": args.noise}
        # (d) the headline run with the fast f32 noise generator
        if args.noise == "f64":
            run(3, noise="f32", n=min(n_global, 65536 * world), steps=2)
            sync_all()
            t0 = time.perf_counter()
            spf, postf = run(4, noise="f32")
            sync_all()
            tf_ = time.perf_counter() - t0
            extra["flow_run_f32_noise"] = {"wall_s": round(tf_, 4), "temperatures": len(spf.history.beta),
                                           "particle_steps_per_s": n_global * len(spf.history.beta) * n_mc / tf_,
                                           "abs_err_in_sigma": abs(float(postf.log_evidence) - true_logz) / max(float(postf.log_evidence_error), 1e-300)}
        # (d2) the same workload with the reference's DEFAULT flow class as the proposal (ZukoFlow(flow_class="MAF"),
        #      flows/torch/flows.py:140-168; here MAFFlow: 3 masked autoregressive transforms, MLP 32 -> 64 -> 64 -> 64): proposal draw
        #      in k_maf_sample, every mutation step in the MAF instantiation of k_pcn_flow_fused
        try:
            from aspire_amd.flows import MAFFlow

            mflow = MAFFlow(d, n_transforms=3, hidden_features=(64, 64), device=eng.device, dtype=torch.float32, seed=1234)
            mflow.fit(sigma_q * 0.9 * np.random.default_rng(3).normal(size=(8000, d)), n_epochs=8)  # untimed
            if sharded:
                mflow.sync_shards(comm)
            run(3, flow=mflow, n=min(n_global, 65536 * world), steps=2)
            sync_all()
            eng.profile(True)
            t0 = time.perf_counter()
            spq, postq = run(4, flow=mflow)
            sync_all()
            tq = time.perf_counter() - t0
            kq = eng.profile_report()
            eng.profile(False)
            fk = next((k for k in kq if k.startswith("k_pcn_flow_fused")), None)
            extra["flow_run_maf"] = {"wall_s": round(tq, 4), "temperatures": len(spq.history.beta),
                                     "particle_steps_per_s": n_global * len(spq.history.beta) * n_mc / tq,
                                     "log_evidence": float(postq.log_evidence),
                                     "abs_err_in_sigma": abs(float(postq.log_evidence) - true_logz) / max(float(postq.log_evidence_error), 1e-300),
                                     "mean_accept": float(np.mean(spq.history.mcmc_acceptance)),
                                     "fused_step_us": round(kq[fk][1] * 1e3, 2) if fk else None,
                                     "draw_us": round(kq["k_maf_sample"][1] * 1e3, 1) if "k_maf_sample" in kq else None,
                                     "torch_ops_in_mutation_loop": 0 if fk and not any(k.startswith("k_pcn_flow_propose") for k in kq) else None}
            # (d3) the reference's DEFAULTS end to end: step_fn="tpcn" and n_steps = 5 d mutation steps per temperature
            #      (smc/minipcn.py:46-49, :89-91) with its default flow class as the proposal (flows/torch/flows.py:140) - the
            #      pairing k_pcn_flow_fused<.., MAF> with the Student-t reference is checked against orc_tpcn_flow_step_kind
            n_ref = 5 * d
            run(3, flow=mflow, n=min(n_global, 65536 * world), steps=2, step_fn="tpcn")
            sync_all()
            eng.profile(True)
            t0 = time.perf_counter()
            spr, postr = run(5, flow=mflow, step_fn="tpcn", steps=n_ref)
            sync_all()
            tr_ = time.perf_counter() - t0
            kr = eng.profile_report()
            eng.profile(False)
            fkr = next((k for k in kr if k.startswith("k_pcn_flow_fused")), None)
            extra["reference_defaults_run"] = {
                "what": f"sample() with the reference's defaults: step_fn='tpcn', n_steps = 5 d = {n_ref} per temperature "
                        "(smc/minipcn.py:46-49), MAF proposal (flows/torch/flows.py:140)",
                "wall_s": round(tr_, 4), "temperatures": len(spr.history.beta), "mcmc_steps_per_temperature": n_ref,
                "particle_steps_per_s": n_global * len(spr.history.beta) * n_ref / tr_,
                "log_evidence": float(postr.log_evidence),
                "abs_err_in_sigma": abs(float(postr.log_evidence) - true_logz) / max(float(postr.log_evidence_error), 1e-300),
                "mean_accept": float(np.mean(spr.history.mcmc_acceptance)),
                "fused_step_us": round(kr[fkr][1] * 1e3, 2) if fkr else None,
                "fused_launches": int(kr[fkr][0]) if fkr else 0,
                "mutation_path": getattr(spr, "last_mutation_path", None)}
        except Exception as exc:  # an extra leg never costs the line
            extra.setdefault("flow_run_maf", {"error": repr(exc)})
            extra.setdefault("reference_defaults_run", {"error": repr(exc)})
        # (d4) flow-proposal steps ABOVE 32 dimensions (round 5, csrc/asmc_flow16.hip): the one-kernel step on 16-particle groups
        #      with streamed weights, coupling and autoregressive proposals at d = 64 / 128 (narrower problems run zero-padded on
        #      these: d = 48 costs what d = 64 does), 8 steps at 1M particles; hbm_frac = (2 d s + 16) bytes per particle / time / peak
        try:
            from aspire_amd.flows import MAFFlow as _MAF

            big = {}
            if n_global * 128 * 8 * 3 < 40e9:  # (state + padded copies)
                eng_big = eng if eng.d_max >= 128 else HipEngine(local_rank, n_max=n_local, d_max=128)
                for kind, dd in (("coupling", 64), ("maf", 64), ("coupling", 128), ("maf", 128)):
                    fl = (CouplingFlow(dd, n_layers=4, hidden_features=(64, 64), device=eng_big.device, dtype=torch.float32, seed=5) if kind == "coupling"
                          else _MAF(dd, n_transforms=3, hidden_features=(64, 64), device=eng_big.device, dtype=torch.float32, seed=5))
                    fl.fit(1.2 * np.random.default_rng(3).normal(size=(4000, dd)), n_epochs=2)  # untimed: non-trivial weights
                    dev_f = fl.device_coupling(eng_big)
                    gx = torch.Generator(eng_big.device).manual_seed(dd)
                    xs_ = torch.randn((n_local, dd), device=eng_big.device, dtype=torch.float64, generator=gx)
                    t_l = eng_big.make_mixture([0.0], np.zeros((1, dd)), np.ones((1, dd)))
                    mu_ = eng_big.asarray(np.zeros(dd))
                    L_ = eng_big.asarray(np.eye(dd))
                    ll_, lp_, lq_ = eng_big.mixture_logpdf(xs_, t_l), eng_big.mixture_logpdf(xs_, t_l), eng_big.coupling_logprob(xs_, dev_f)
                    for nu_, name in ((0.0, "pcn"), (5.0, "tpcn")):
                        args_ = (xs_, ll_, lp_, lq_, 0.5, mu_, L_, L_, t_l, t_l, dev_f, 7, 0, 0.15, 8, 0, 0.234, False, "f64", nu_)
                        eng_big.pcn_mutate_flow(*args_)
                        eng_big.profile(True)
                        eng_big.pcn_mutate_flow(*args_)
                        kk = eng_big.profile_report()
                        eng_big.profile(False)
                        sk = next((k for k in kk if k.startswith(("k_pcn_flow16", "k_tpcn_flow16"))), None)
                        ms_ = kk[sk][1] if sk else None
                        big[f"{kind}_d{dd}_{name}"] = {
                            "step_kernel": sk, "us_per_step": round(ms_ * 1e3, 1) if ms_ else None,
                            "hbm_frac": round((2 * dd * 8 + 16) * n_local / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_ else None,
                            "kernels_per_step": round(sum(c for k, (c, _) in kk.items() if k.startswith(("k_pcn", "k_tpcn", "k_flow", "k_coupling", "k_mixture", "k_copy", "k_gamma")) and "whiten" not in k and "adapt" not in k) / 8, 2)}
                    del xs_, ll_, lp_, lq_
                if eng_big is not eng:
                    eng_big.close()
            extra["flow_step_above_32_dims"] = big
        except Exception as exc:
            extra["flow_step_above_32_dims"] = {"error": repr(exc)}
        # (d5) BASELINE configs[4]'s mutation step on ONE GPU: 1M x 128, two-component mixture likelihood, analytic proposal,
        #      k_pcn_mm on the fp64 matrix cores (8 pCN steps, default noise); algorithmic bytes 2 d s + 16 = 2064 per particle
        try:
            d5 = 128
            eng5 = eng if eng.d_max >= d5 else HipEngine(local_rank, n_max=n_local, d_max=d5)
            lik5 = DiagGaussianMixture(np.stack([2 * np.ones(d5), -2 * np.ones(d5)]), np.stack([0.5 * np.ones(d5), np.ones(d5)]))
            pri5 = DiagGaussianMixture.isotropic(d5, 0.0, 1.0)
            q5 = GaussianFlow(d5, sigma=3.0, engine=eng5, seed=4, dtype=xdt)
            x5, lq5 = q5.sample_and_log_prob(n_local)
            ll5 = eng5.mixture_logpdf(x5, lik5.device_mixture(eng5))
            lp5 = eng5.mixture_logpdf(x5, pri5.device_mixture(eng5))
            mu5, L5 = eng5.asarray(np.zeros(d5)), eng5.asarray(3.0 * np.eye(d5))
            Li5 = eng5.asarray(np.eye(d5) / 3.0)
            c5 = {}
            for nu_, name in ((0.0, "pcn"), (5.0, "tpcn")):
                a5 = (x5, ll5, lp5, lq5, 0.3, mu5, L5, Li5, lik5.device_mixture(eng5), pri5.device_mixture(eng5), q5.device_mixture(eng5), 7, 0,
                      0.1, 8, 0, 0.234, False)
                eng5.pcn_mutate(*a5, noise=args.noise, nu=nu_)
                eng5.profile(True)
                eng5.pcn_mutate(*a5, noise=args.noise, nu=nu_)
                k5 = eng5.profile_report()
                eng5.profile(False)
                sk = next((k for k in k5 if k.startswith(("k_pcn_mm_step", "k_tpcn_mm_step"))), None)
                ms_ = k5[sk][1] if sk else None
                c5[name] = {"step_kernel": sk, "us_per_step": round(ms_ * 1e3, 1) if ms_ else None,
                            "algorithmic_bytes_per_particle": 2 * d5 * s_bytes + 16,
                            "hbm_frac": round((2 * d5 * s_bytes + 16) * n_local / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_ else None,
                            "fp64_matrix_flops_per_particle": 144 * 2048 // 16}
            extra["config5_step"] = c5
            if eng5 is not eng:
                eng5.close()
        except Exception as exc:
            extra["config5_step"] = {"error": repr(exc)}
        # (d6) BASELINE configs[4] END TO END on one GPU: 1M x 128, equal-weight mixture of two Gaussians
        #      (mu = +-2, cov 1/2 I and I: examples/smc_example.py:34-53 lifted to d = 128), prior N(0, I), q = N(0, 3^2 I), adaptive
        #      tempering, the reference's default step (tpCN), n_mc steps per temperature, default noise; log Z against the closed form
        try:
            d5 = 128
            eng5 = eng if eng.d_max >= d5 else HipEngine(local_rank, n_max=n_local, d_max=d5)
            lik5 = DiagGaussianMixture(np.stack([2 * np.ones(d5), -2 * np.ones(d5)]), np.stack([0.5 * np.ones(d5), np.ones(d5)]))
            pri5 = DiagGaussianMixture.isotropic(d5, 0.0, 1.0)

            def lg5(mu_, var_):
                return -0.5 * d5 * np.log(2 * np.pi * var_) - 0.5 * d5 * mu_ * mu_ / var_

            true5 = float(np.logaddexp(np.log(0.5) + lg5(2.0, 1.5), np.log(0.5) + lg5(2.0, 2.0)))

            def run5(nn, seed_):
                sp5 = HipSMC(log_likelihood=lik5, log_prior=pri5, dims=d5, prior_flow=GaussianFlow(d5, sigma=3.0, engine=eng5, seed=4, dtype=xdt),
                             xp=np, engine=eng5, rng=np.random.default_rng(seed_), dtype=xname)
                return sp5, sp5.sample(nn, sampler_kwargs=dict(n_steps=n_mc, step_fn="tpcn", noise=args.noise), store_sample_history=False)

            run5(65536, 1)
            torch.cuda.synchronize()
            eng5.profile(True)
            t0 = time.perf_counter()
            sp5, post5 = run5(n_local, 2)
            torch.cuda.synchronize()
            t5 = time.perf_counter() - t0
            k5r = eng5.profile_report()
            eng5.profile(False)
            nt5 = len(sp5.history.beta)
            sk5 = next((k for k in k5r if k.startswith(("k_tpcn_mm_step", "k_pcn_mm_step"))), None)
            ms5 = k5r[sk5][1] if sk5 else None
            busy5 = sum(c * ms for c, ms in k5r.values())
            conv5 = {k: round(c * ms / nt5, 4) for k, (c, ms) in k5r.items() if "whiten" in k}
            extra["config5_run"] = {
                "workload": f"configs[4] on one GPU: {n_local} x {d5}, two-component mixture likelihood, N(0, I) prior, q = N(0, 9 I), adaptive "
                            f"tempering, tpCN, {n_mc} steps per temperature, {args.noise} noise",
                "wall_s": round(t5, 4), "temperatures": nt5, "particle_steps_per_s": n_local * nt5 * n_mc / t5,
                "log_evidence": float(post5.log_evidence), "log_evidence_closed_form": true5,
                "log_evidence_error": float(post5.log_evidence_error),
                "abs_err_in_sigma": abs(float(post5.log_evidence) - true5) / max(float(post5.log_evidence_error), 1e-300),
                "mean_accept": float(np.mean(sp5.history.mcmc_acceptance)),
                "step_kernel": sk5, "step_us": round(ms5 * 1e3, 1) if ms5 else None,
                "step_launches": int(k5r[sk5][0]) if sk5 else 0,
                "algorithmic_bytes_per_particle_step": 2 * d5 * s_bytes + 16,
                "hbm_frac": round((2 * d5 * s_bytes + 16) * n_local / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms5 else None,
                "gpu_busy_over_wall": round(busy5 * 1e-3 / t5, 4),
                "state_conversion_ms_per_temperature": conv5,
                "mutation_path": getattr(sp5, "last_mutation_path", None)}
            if eng5 is not eng:
                eng5.close()
        except Exception as exc:
            extra["config5_run"] = {"error": repr(exc)}
        # (d7) the headline workload with fp32 STATE ("x fp64 and fp32 variants"; log-probabilities stay fp64, H7):
        #      algorithmic bytes per particle-step 2 d 4 + 16 = 272
        try:
            other = "float32" if xname == "float64" else "float64"
            run(3, n=min(n_global, 65536 * world), steps=2, x_dtype=other)
            sync_all()
            eng.profile(True)
            t0 = time.perf_counter()
            sp32, post32 = run(4, x_dtype=other)
            sync_all()
            t32 = time.perf_counter() - t0
            k32 = eng.profile_report()
            eng.profile(False)
            fk32 = next((k for k in k32 if k.startswith("k_pcn_flow_fused")), None)
            b32 = (2 * d * (4 if other == "float32" else 8) + 16) * n_local
            extra["headline_x_f32" if other == "float32" else "headline_x_f64"] = {
                "x_dtype": other, "wall_s": round(t32, 4), "temperatures": len(sp32.history.beta),
                "particle_steps_per_s": n_global * len(sp32.history.beta) * n_mc / t32,
                "log_evidence": float(post32.log_evidence),
                "abs_err_in_sigma": abs(float(post32.log_evidence) - true_logz) / max(float(post32.log_evidence_error), 1e-300),
                "mean_accept": float(np.mean(sp32.history.mcmc_acceptance)),
                "step_kernel": fk32, "step_us": round(k32[fk32][1] * 1e3, 2) if fk32 else None,
                "algorithmic_bytes_per_launch": b32,
                "hbm_frac": round(b32 / (k32[fk32][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fk32 else None,
                "returned_x_dtype": str(getattr(post32.x, "dtype", None))}
        except Exception as exc:
            extra["headline_x_f32"] = {"error": repr(exc)}
        # (d8) the headline workload with Python CALLABLES as the densities (the path every user of the reference's seam takes:
        #      smc/base.py:507-519 calls log_prior, then log_likelihood, on a Samples-like object each step): torch callables on the
        #      device rows, the trained coupling flow as the proposal; the library's launches per step are counted from its own
        #      events, the callables' torch ops ride between them
        try:
            def t_like(smp):
                return -0.5 * (smp.x * smp.x).sum(1)

            run(3, n=min(n_global, 65536 * world), steps=2, targets=(t_like, t_like), xp_=torch)
            sync_all()
            eng.profile(True)
            t0 = time.perf_counter()
            spc, postc = run(4, targets=(t_like, t_like), xp_=torch)
            sync_all()
            tc_ = time.perf_counter() - t0
            kc = eng.profile_report()
            eng.profile(False)
            ntc = len(spc.history.beta)
            lib_launches = sum(c for c, _ in kc.values())
            lev = postc.log_evidence
            extra["callables_run"] = {
                "what": "configs[2] with torch-callable log_likelihood / log_prior (split path: propose kernel -> flow density -> callables -> accept kernel)",
                "wall_s": round(tc_, 4), "temperatures": ntc, "particle_steps_per_s": n_global * ntc * n_mc / tc_,
                "ms_per_step": round(tc_ / (ntc * n_mc) * 1e3, 4),
                "library_launches_per_step": round(lib_launches / (ntc * n_mc), 2),
                "library_gpu_ms_per_step": round(sum(c * ms for c, ms in kc.values()) / (ntc * n_mc), 4),
                "log_evidence": float(lev), "abs_err_in_sigma": abs(float(lev) - true_logz) / max(float(postc.log_evidence_error), 1e-300),
                "mean_accept": float(np.mean(spc.history.mcmc_acceptance)),
                "likelihood_evaluations": int(spc.n_likelihood_evaluations),
                "mutation_path": getattr(spc, "last_mutation_path", None),
                "top_kernels_us": {k: [int(c), round(ms * 1e3, 1)] for k, (c, ms) in sorted(kc.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:6]}}
        except Exception as exc:
            extra["callables_run"] = {"error": repr(exc)}
        # (e) the headline run with the flow on the fp32 MFMA chain (v_mfma_f32_32x32x2_f32) instead of the default split-fp16
        #     products: same operands to fp32 accuracy, 16/3 of the matrix-pipe time
        if flow_math == "f16x2-split":
            os.environ["ASMC_FLOW_MATH"] = "f32"
            try:
                run(3, n=min(n_global, 65536 * world), steps=2)
                sync_all()
                t0 = time.perf_counter()
                spm, postm = run(4)
                sync_all()
                tm_ = time.perf_counter() - t0
                extra["flow_run_f32_mfma"] = {"wall_s": round(tm_, 4), "temperatures": len(spm.history.beta),
                                              "particle_steps_per_s": n_global * len(spm.history.beta) * n_mc / tm_,
                                              "log_evidence": float(postm.log_evidence),
                                              "abs_err_in_sigma": abs(float(postm.log_evidence) - true_logz) / max(float(postm.log_evidence_error), 1e-300)}
            finally:
                os.environ.pop("ASMC_FLOW_MATH", None)
            # top level next to `value`: the same workload with every flow product on the fp32-input matrix instruction
            result["value_strict_fp32"] = extra["flow_run_f32_mfma"]["particle_steps_per_s"]
        # (f) what the SHARDED code path costs before any wire time: the headline run through the sharded
        #     machinery over a ONE-rank RCCL group made in this process - every collective of the hot path is issued (the beta
        #     search's records, the evidence partials, tile records and chain states of the exact cdf, offspring counts, the
        #     per-step accept counts, the reference fit's moments), none has a peer - against the single-rank runs, interleaved
        if not sharded and not os.environ.get("ASMC_BENCH_NO_SHARDED_LEG"):
            try:
                import torch.distributed as dist

                from aspire_amd.comm import TorchDistComm

                made_group = False
                if not dist.is_initialized():
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
                    made_group = True
                scomm = TorchDistComm(eng.device)
                scomm.force_sharded = True
                run(5, n=65536, steps=2, comm_=scomm)
                run(6, comm_=scomm)
                sync_all()
                t_single, t_shard = [], []
                for k in range(4):
                    for which, acc in ((None, t_single), (scomm, t_shard)):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        sps, posts = run(2000 + k, comm_=which)
                        torch.cuda.synchronize()
                        acc.append((time.perf_counter() - t0) * 1e3)
                eng.profile(True)
                sps, posts = run(2100, comm_=scomm)
                torch.cuda.synchronize()
                ks = eng.profile_report()
                eng.profile(False)
                ms1, ms2 = float(np.median(t_single)), float(np.median(t_shard))
                extra["sharded_path_one_rank_group"] = {
                    "what": "sample() through the sharded code path over a one-rank RCCL group (all collectives issued, no peer) "
                            "against the single-rank path, 4 runs each, interleaved, same workload as the headline",
                    "ms_per_run_single": round(ms1, 3), "ms_per_run_sharded": round(ms2, 3), "overhead": round(ms2 / ms1 - 1.0, 4),
                    "importance_step_as_one_chain": bool(smc_math.shard_step_available(eng, scomm)
                                                         and "k_weights_m2_lse_shard" in ks and "k_bis_decide" not in ks),
                    "host_synchronisations_per_temperature_in_the_step": 1,
                    "abs_err_in_sigma": abs(float(posts.log_evidence) - true_logz) / max(float(posts.log_evidence_error), 1e-300),
                    "library_issues_its_own_collectives": bool(scomm.rccl_direct() is not None)}
                if made_group:
                    dist.destroy_process_group()
            except Exception as exc:  # an extra leg never costs the line
                extra["sharded_path_one_rank_group"] = {"error": repr(exc)}
        result["extra"] = extra

    # ---- CPU baseline: the oracle's restatement of the SAME mutation step (kind "port") on the host's cores -------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O

        ws, bs = cflow.export_layers()
        loc, scale = cflow.loc.detach().cpu().numpy(), cflow.scale.detach().cpu().numpy()
        g = np.random.default_rng(5)
        n_cpu = min(n_local, 1_000_000)
        xc = np.sqrt(0.6) * g.normal(size=(n_cpu, d))  # a mid-schedule population
        tg = O.Mixture([0.0], np.zeros((1, d)), np.ones((1, d)))
        llc = -0.5 * np.sum(xc * xc, axis=1)
        lpc = llc.copy()
        lqc = np.zeros(n_cpu)
        mu_c, L_c = np.zeros(d), np.sqrt(0.6) * np.eye(d)
        Li_c = np.eye(d) / np.sqrt(0.6)
        n_thr = O.max_threads()

        def cpu_steps(n_part, threads, budget_s, max_steps):
            xs, a, b, c = xc[:n_part].copy(), llc[:n_part].copy(), lpc[:n_part].copy(), lqc[:n_part].copy()
            O.pcn_flow_step(xs[:256].copy(), a[:256].copy(), b[:256].copy(), c[:256].copy(), 0.5, mu_c, L_c, Li_c, 0.5, tg, tg, ws, bs,
                            loc, scale, 9, 0, 0, "f64", threads)
            t0, k = time.perf_counter(), 0
            while True:
                O.pcn_flow_step(xs, a, b, c, 0.5, mu_c, L_c, Li_c, 0.5, tg, tg, ws, bs, loc, scale, 9, 0, k, "f64", threads)
                k += 1
                el = time.perf_counter() - t0
                if el > budget_s or k >= max_steps:
                    return n_part * k / el, k, el

        v1, k1, e1 = cpu_steps(min(n_cpu, 32768), 1, 5.0, 8)
        # thread sweep on bounded samples (a container may own fewer cores than os.cpu_count() shows: the cgroup quota is
        # reported, and the best point of the sweep is the baseline), then the longer run at the best thread count
        def cgroup_cpus():
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                return None if q == "max" else float(q) / float(per)
            except Exception:
                return None

        def affinity_cpus():
            try:
                return len(os.sched_getaffinity(0))
            except Exception:
                return None

        sweep = {}
        for thr in sorted({t for t in (8, 32, n_thr // 2, n_thr) if 1 < t <= n_thr}):
            vs_, ks_, es_ = cpu_steps(min(n_cpu, 4096 * thr), thr, 2.5, 6)
            sweep[thr] = vs_
        best_thr = max(sweep, key=sweep.get) if sweep else 1
        vN, kN, eN = cpu_steps(n_cpu, best_thr, 10.0, 64)
        result["cpu_baseline"] = {
            "value": vN, "unit": "particle-steps/s", "cores": best_thr, "kind": "port",
            "sample": f"{kN} mutation steps (pCN propose + coupling-flow log q in fp32 + targets + accept, the same flow and "
                      f"noise streams; oracle/asmc_oracle.c orc_pcn_flow_step, OpenMP over particles, {best_thr} threads = the best "
                      f"point of the thread sweep) on {n_cpu} x {d} fp64 particles, {eN:.1f} s",
            "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "omp_max_threads": n_thr,
            "cgroup_cpu_quota": cgroup_cpus(), "affinity_cpus": affinity_cpus(),
            "single_thread": {"value": v1, "cores": 1, "sample": f"{k1} steps on {min(n_cpu, 32768)} particles, {e1:.1f} s"},
            "thread_sweep": {str(t): round(v) for t, v in sweep.items()},
            "speedup_over_single_thread": round(vN / v1, 1),
            # efficiency against the cores the container may actually use: the cgroup quota caps it (the GPU boxes of this pool
            # grant 16 CPUs of a 256-thread host: 128 OpenMP threads time-share them, which is why round 2's "128 cores" scaled 12x)
            "effective_cores": (min(best_thr, cgroup_cpus()) if cgroup_cpus() else best_thr),
            "parallel_efficiency": round(vN / v1 / (min(best_thr, cgroup_cpus()) if cgroup_cpus() else best_thr), 3),
            "gpu_over_cpu_all_cores": value / vN, "gpu_over_cpu_single_thread": value / v1,
        }
        try:  # restatement-to-reference ratio, measured in the build container (tests/tools/ref_ratio.py)
            result["cpu_baseline"]["port_vs_reference"] = json.load(open(os.path.join(ROOT, "profiles", "ref_ratio.json")))
        except Exception:
            pass
    # The JSON line must be the LAST thing on the job's stdout.  RCCL writes a version banner through C stdio, which is
    # flushed when a process exits - after Python's own prints: every rank flushes it now, ranks other than 0 close
    # their stdout before the barrier, and rank 0 closes its own right after the line.
    def hush():
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)

    if sharded:
        if rank != 0:
            hush()
        comm.barrier()
    if rank == 0:
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), file=record_out, flush=True)
        hush()
    if sharded:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
