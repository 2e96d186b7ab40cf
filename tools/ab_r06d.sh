#!/bin/bash
# same-box A/B of the round-6 late kernel changes (noise loop without per-quad dimension tests, vector bias / row-table reads, unconditional
# row loads) against the build of the commit before them: tools/ab_r06d.sh <out-prefix> <base.so>
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
P=$1; B=$2
tools/ab_flowstep_maf.sh ${P}_fused.txt $B tree
tools/ab_flow16.sh ${P}_flow16.txt $B tree
: > ${P}_config5_step.txt
for rep in 1 2 3; do
  for v in $B tree; do
    if [ "$v" = tree ]; then r=$(NOISE=f64 STEPS=8 TOP=4 python tools/config5.py 2>&1 | grep -E "mm_step|particle-steps" | tr -s " " | tr '\n' ' '); else r=$(NOISE=f64 STEPS=8 TOP=4 ASMC_LIB_PATH=$v python tools/config5.py 2>&1 | grep -E "mm_step|particle-steps" | tr -s " " | tr '\n' ' '); fi
    echo "$(basename $v): $r" | tee -a ${P}_config5_step.txt
  done
done
