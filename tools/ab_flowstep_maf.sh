#!/bin/bash
# A/B of library builds on the d = 32 flow-proposal step with an autoregressive proposal (KIND=maf): tools/ab_flowstep_maf.sh <out.txt> tree|variant.so ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; shift
: > $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    for kind in ${KINDS:-maf coupling}; do
      if [ "$v" = tree ]; then
        r=$(KIND=$kind RHO=0.02 ADAPT=0 STEPS=32 python tools/flowstep_bench.py 2>&1 | grep -E "ms/step|k_pcn_flow_fused" | tr -s ' ' | tr '\n' ' ')
      else
        r=$(KIND=$kind RHO=0.02 ADAPT=0 STEPS=32 ASMC_LIB_PATH=$v python tools/flowstep_bench.py 2>&1 | grep -E "ms/step|k_pcn_flow_fused" | tr -s ' ' | tr '\n' ' ')
      fi
      echo "$(basename $v) $kind: $r" | tee -a $OUT
    done
  done
done
