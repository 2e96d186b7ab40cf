# One GPU, one rank: the run of bench.py through the single-rank path and through the SHARDED path (RCCL group of one
# rank: every collective of the hot path is issued, none has a peer) - the difference is the cost of the sharded
# machinery itself (extra launches, host synchronisations, the collectives' launch latency), i.e. an upper bound on the
# weak-scaling efficiency before any wire time.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
python bench.py --gpus 1 --steps ${RIG_STEPS:-10} --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/rig1_single.json 2> gpurun_out/rig1_single.err
python bench.py --gpus 1 --steps ${RIG_STEPS:-10} --warmup 3 --no-extra --no-cpu-baseline --force-sharded > gpurun_out/rig1_sharded.json 2> gpurun_out/rig1_sharded.err
tail -3 gpurun_out/rig1_sharded.err
python - <<'P'
import json
for k in ("single", "sharded"):
    try:
        r = json.loads(open(f"gpurun_out/rig1_{k}.json").read().strip().splitlines()[-1])
        print(k, "value %.4g" % r["value"], "ms/run %.2f" % r["ms_per_step"], r["config"].get("parallelism"))
    except Exception as e:
        print(k, "failed", e)
P
