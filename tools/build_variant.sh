#!/bin/bash
# builds libasmc_hip.so with extra flags for ONE translation unit (diagnostic variants): tools/build_variant.sh asmc_pcn_fused.hip "-DFUSED_STAMP" out.so
set -e
cd "$(dirname "$0")/../aspire_amd/csrc"
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function $2 -c $1 -o /tmp/variant.o
objs=""
for f in _obj/*.o; do b=$(basename $f .o); if [ "$b.hip" != "$1" ]; then objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $3 $objs /tmp/variant.o
