#!/bin/bash
# rocprofv3 kernel stats of the bounded-prior mutation path (tools/boundedprof.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/trace_bounded -o b --output-format csv -- python3 $R/tools/boundedprof.py > $R/gpurun_out/bounded_under_rocprof.log 2>&1
cp /tmp/trace_bounded/b_kernel_stats.csv $R/gpurun_out/bounded_kernel_stats.csv
