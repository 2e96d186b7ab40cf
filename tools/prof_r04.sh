#!/bin/bash
# Round-4 profile on the GPU box (one call): rocprofv3 kernel-trace stats of the DRIVER's bench command, its JSON line, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) of the bench run and of the flow-proposal step at two acceptance rates, and
# the SQ counters of the fused step at both.  Only summaries are kept under gpurun_out/<tag>/.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/trace_$TAG -o bench --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp /tmp/trace_$TAG/bench_kernel_stats.csv $O/bench_kernel_stats.csv
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmcb_$c -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmcb_$c > $O/pmc_bench_$c.txt
  rocprofv3 --pmc $c -d /tmp/pmck_$c -o k --output-format csv -- python3 $R/tools/kbench.py gather pcn cdf weights flow > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmck_$c > $O/pmc_kbench_$c.txt
  for mode in hi mid; do
    if [ $mode = hi ]; then export RHO=0.02 ADAPT=0; else export RHO=0.3 ADAPT=1; fi
    rocprofv3 --pmc $c -d /tmp/pmcf_${mode}_$c -o f --output-format csv -- python3 $R/tools/flowstep_bench.py > $O/flowstep_${mode}.log 2>&1
    python3 $R/tools/pmc_summary.py /tmp/pmcf_${mode}_$c k_pcn_flow_fused > $O/pmc_flowstep_${mode}_$c.txt
  done
done
for mode in hi mid; do
  if [ $mode = hi ]; then export RHO=0.02 ADAPT=0; else export RHO=0.3 ADAPT=1; fi
  TAG=${TAG}_$mode bash $R/tools/pmc_flowstep.sh > /dev/null 2>&1
done
unset RHO ADAPT
du -sh $O; ls $O $R/gpurun_out/pmc_flowstep_${TAG}_hi $R/gpurun_out/pmc_flowstep_${TAG}_mid
