#!/bin/bash
# A/B of library builds on the flow-proposal step micro-driver: tools/ab_flowstep.sh <out.txt> tree|path/to/variant.so ...
# two operating points per build: ~98 % acceptance (RHO=0.02 ADAPT=0, the headline's regime) and adapted to ~23-30 % (RHO=0.3)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; shift
: > $OUT
for v in "$@"; do
  for mode in "RHO=0.02 ADAPT=0" "RHO=0.3 ADAPT=1"; do
    for rep in 1 2; do
      if [ "$v" = tree ]; then
        r=$(env $mode STEPS=32 python tools/flowstep_bench.py 2>&1 | grep -E "ms/step|k_pcn_flow_fused|wave " | tr '\n' ' ')
      else
        r=$(env $mode STEPS=32 ASMC_LIB_PATH=$v python tools/flowstep_bench.py 2>&1 | grep -E "ms/step|k_pcn_flow_fused|wave " | tr '\n' ' ')
      fi
      echo "$(basename $v) [$mode] $r" | tee -a $OUT
    done
  done
done
