#!/bin/bash
# same-box A/B of library builds on the register-resident pCN / tpCN step (tools/pcn_step_bench.py): tools/ab_pcn_step.sh <out.txt> tree|variant.so ...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; shift
: > $OUT
for rep in 1 2 3; do
  for v in "$@"; do
    for nu in 0 5; do
      if [ "$v" = tree ]; then r=$(NU=$nu python tools/pcn_step_bench.py 2>&1 | grep -E "reg_y|accept" | tr -s ' ' | tr '\n' ' ')
      else r=$(NU=$nu ASMC_LIB_PATH=$v python tools/pcn_step_bench.py 2>&1 | grep -E "reg_y|accept" | tr -s ' ' | tr '\n' ' '); fi
      echo "$(basename $v) nu=$nu: $r" | tee -a $OUT
    done
  done
done
