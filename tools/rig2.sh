cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export ASMC_BENCH_BACKEND=gloo ASMC_BENCH_DEVICE=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 2 --no-cpu-baseline --sharded-extras ${RIG_ARGS:-} 2>&1 | grep -v 'hostname of the client\|amdgpu.ids' | head -80
