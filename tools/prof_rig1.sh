#!/bin/bash
# Kernel-trace stats of bench.py through the single-rank path and through the sharded path over a one-rank RCCL group
# (see tools/rig1.sh): which launches the sharded machinery adds, and what they cost.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rig1
mkdir -p $O
for k in single sharded; do
  F=""; [ $k = sharded ] && F="--force-sharded"
  rocprofv3 --kernel-trace --stats -d /tmp/trace_$k -o bench --output-format csv -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 5 --warmup 2 $F > $O/$k.log 2>&1
  cp /tmp/trace_$k/bench_kernel_stats.csv $O/${k}_kernel_stats.csv
  TIMELINE=1 python3 $R/tools/gap_trace.py /tmp/trace_$k/bench_kernel_trace.csv > $O/${k}_gaps.txt 2>&1
  ALL=1 TIMELINE=p90 python3 $R/tools/gap_trace.py /tmp/trace_$k/bench_kernel_trace.csv > $O/${k}_gaps_p90.txt 2>&1
done
[ -z "$NO_RIG" ] && bash $R/tools/rig1.sh
cat $O/*_gaps.txt
