#!/bin/bash
# quick A/B at the headline's operating point only (RHO=0.02 ADAPT=0): tools/ab_quick.sh <out.txt> tree|variant.so ...  (two passes)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; shift
: > $OUT
for pass in 1 2; do
for v in "$@"; do
  if [ "$v" = tree ]; then
    r=$(RHO=0.02 ADAPT=0 STEPS=32 python tools/flowstep_bench.py 2>&1 | grep -E "k_pcn_flow_fused" | tr '\n' ' ')
  else
    r=$(RHO=0.02 ADAPT=0 STEPS=32 ASMC_LIB_PATH=$v python tools/flowstep_bench.py 2>&1 | grep -E "k_pcn_flow_fused" | tr '\n' ' ')
  fi
  echo "$(basename $v) $r" | tee -a $OUT
done
done
