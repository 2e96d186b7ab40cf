#!/bin/bash
# two 4-wave blocks per CU with a start skew (diagnostic builds -DF16_SKEW=k) against the tree's one 8-wave block: tools/ab_skew.sh <out.txt> <so>...
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; shift
: > $OUT
for rep in 1 2; do
  r=$(KIND=coupling D=64 python tools/flow16_bench.py 2>&1 | grep -E "flow16 " | tr -s ' ' | tr '\n' ' '); echo "tree(whole, 8 waves): $r" | tee -a $OUT
  r=$(ASMC_F16_SMALL_SLOTS=1 KIND=coupling D=64 python tools/flow16_bench.py 2>&1 | grep -E "flow16 " | tr -s ' ' | tr '\n' ' '); echo "tree(32 KB slots, 8 waves): $r" | tee -a $OUT
  for v in "$@"; do
    r=$(KIND=coupling D=64 ASMC_LIB_PATH=$v python tools/flow16_bench.py 2>&1 | grep -E "flow16 " | tr -s ' ' | tr '\n' ' '); echo "$(basename $v): $r" | tee -a $OUT
  done
done
