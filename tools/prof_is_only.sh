#!/bin/bash
# rocprofv3 kernel stats of the IS-only bench legs alone (no extra legs, no CPU baseline): every k_gather16 launch in the
# trace is a 1M x 32 step of the timed workload, so its average is directly comparable with roofline.avg_ms.
TAG=${1:-r01l}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/trace_is_$TAG -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra > $O/bench_is_only_under_rocprof.log 2>&1
cp /tmp/trace_is_$TAG/bench_kernel_stats.csv $O/bench_is_only_kernel_stats.csv
grep -h '^{' $O/bench_is_only_under_rocprof.log | tail -1 > $O/bench_is_only.json
head -12 $O/bench_is_only_kernel_stats.csv | cut -c1-150
