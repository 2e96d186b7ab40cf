# A/B of two builds of the library on one box through the headline run: ab_bench.sh <variant.so> [reps]
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
V=$1; R=${2:-3}
for i in $(seq $R); do
  a=$(python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/run, fused %.1f us' % (r['ms_per_step'], 1e3*r['roofline']['avg_ms']))")
  b=$(ASMC_LIB_PATH=$V python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/run, fused %.1f us' % (r['ms_per_step'], 1e3*r['roofline']['avg_ms']))")
  echo "tree: $a | variant: $b"
done
