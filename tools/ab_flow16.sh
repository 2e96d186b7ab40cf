#!/bin/bash
# A/B of library builds on the flow16 step micro-driver (tools/flow16_bench.py): tools/ab_flow16.sh <out.txt> tree|path/to/variant.so ...
# four shapes (coupling / maf x D = 64 / 128), pCN, 8 steps at 1M particles, two repetitions per build, builds interleaved
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=$1; shift
: > $OUT
for rep in 1 2; do
  for v in "$@"; do
    for shape in "coupling 64" "maf 64" "coupling 128" "maf 128"; do
      kind=${shape% *}; dd=${shape#* }
      if [ "$v" = tree ]; then
        r=$(KIND=$kind D=$dd NU=${NU:-0} python tools/flow16_bench.py 2>&1 | grep -E "flow16 " | tr -s ' ' | tr '\n' ' ')
      else
        r=$(KIND=$kind D=$dd NU=${NU:-0} ASMC_LIB_PATH=$v python tools/flow16_bench.py 2>&1 | grep -E "flow16 " | tr -s ' ' | tr '\n' ' ')
      fi
      echo "$(basename $v) $kind d=$dd: $r" | tee -a $OUT
    done
  done
done
