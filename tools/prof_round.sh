#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats of bench.py, the bench JSON line, and HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs).  Only summaries are kept under gpurun_out/<tag>/ (raw traces are large).
TAG=${1:-r01c}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/trace_$TAG -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp /tmp/trace_$TAG/bench_kernel_stats.csv $O/bench_kernel_stats.csv
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmc_$c -o k --output-format csv -- python3 $R/tools/kbench.py gather pcn cdf weights flow > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmc_$c > $O/pmc_kbench_$c.txt
  rocprofv3 --pmc $c -d /tmp/pmcb_$c -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmcb_$c > $O/pmc_bench_$c.txt
done
du -sh $O; ls $O
