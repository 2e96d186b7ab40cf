"""The driver's contract with bench.py, checked on a small workload: ONE JSON line on stdout carrying the metric BASELINE.json
names, whole-job throughput, the roofline of the dominant kernel measured in the run, the CPU baseline timed beside it, and the
extra legs - none of which may fail silently (a leg that raised reports {"error": ...})."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_cpu_baseline_and_extra_legs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--particles-per-gpu", "262144"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    j = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"].split(";")[0].replace(" x ", "×").replace("N×n_steps", "N×n_steps")[:14] == base["metric"][:14]
    assert j["unit"] == "particle-steps/s" and j["value"] > 0 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["ms_per_step"] > 0 and j["vs_baseline"] is None
    assert j["dtype"] == "f64" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    assert rf["kernel"] == "k_pcn_flow_fused" and 0 < rf["frac_algorithmic"] < rf["frac"] < 1 and rf["avg_ms"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"] and cb["unit"] == j["unit"]
    legs = j["extra"]
    for name in ("is_only_step", "fused_step_by_regime", "flow_run_maf", "flow_run_f32_mfma", "sharded_path_one_rank_group",
                 "config5_run", "headline_x_f32", "callables_run"):
        assert name in legs and "error" not in legs[name], (name, legs.get(name))
    # round 6 legs: configs[4] end to end (tpCN, d = 128), the headline with fp32 state, the headline with callable densities
    c5 = legs["config5_run"]
    assert c5["step_kernel"].startswith("k_tpcn_mm_step") and c5["temperatures"] > 5 and c5["abs_err_in_sigma"] < 5 and 0 < c5["hbm_frac"] < 1
    f32 = legs["headline_x_f32"]
    assert f32["x_dtype"] == "float32" and f32["step_kernel"].startswith("k_pcn_flow_fused") and "float32" in f32["returned_x_dtype"] and f32["abs_err_in_sigma"] < 5
    cal = legs["callables_run"]
    assert cal["ms_per_step"] > 0 and cal["library_launches_per_step"] >= 2 and "callables" in cal["mutation_path"] and cal["abs_err_in_sigma"] < 5
    lim = rf["limiter"]
    assert isinstance(lim, dict) and lim["name"] in lim["shares"] and lim["shares"][lim["name"]] == max(lim["shares"].values())
    if lim.get("counters_current"):  # the committed counters belong to this tree's kernel sources: the line carries them
        assert rf["traffic"] and 0.9 < rf["traffic_over_algorithmic"] < 1.15  # no wasted HBM traffic in the dominant kernel
        assert rf["valu_insts_per_64_particle_tile"] <= 4400  # VERDICT r5 item 6 (round 5: 4 998)
    assert legs["sharded_path_one_rank_group"]["importance_step_as_one_chain"] is True
    assert legs["flow_run_maf"]["torch_ops_in_mutation_loop"] == 0


def test_bench_gpus_2_launches_its_own_ranks_and_reports_both():
    """`python bench.py --gpus 2` as the driver runs `--gpus 1` - no launcher around it: the process starts its own two ranks
    before any GPU call and relays rank 0's line.  One-GPU rig: both ranks on GPU 0, collectives staged
    through gloo.  The line must describe the TWO-rank job: n_gpus 2, global population 2 x per-GPU (samples.py:1277-1278 acts on
    the global population)."""
    env = dict(os.environ, ASMC_BENCH_BACKEND="gloo", ASMC_BENCH_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--particles-per-gpu", "131072", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    cfg = j["config"]
    assert cfg["particles_per_gpu"] == 131072 and cfg["global_particles"] == 2 * cfg["particles_per_gpu"]
    assert cfg["parallelism"].startswith("particle-shard x2")


def test_bench_refuses_more_gpus_than_are_visible():
    """... and when the ranks cannot be started it exits non-zero instead of measuring one GPU."""
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ASMC_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == "" and "refusing" in r.stderr
