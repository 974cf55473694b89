#!/bin/bash
# Build a variant of ONE OR MORE kernel families for A/B runs: the listed plasma translation units are recompiled with
# extra flags into tps_amd/csrc/_ab/<name>/libtpsrhs_<unit>.so; run with TPSRHS_FAMILY_PATH=$PWD/tps_amd/csrc/_ab/<name>
# (the core library looks there first, every other family comes from the regular build).
#   tools/build_variant.sh <name> "<flags>" unit1 [unit2 ...]
# A unit may be given as _probe/<file> (a development translation unit exporting the same pick_<unit> entry).
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../tps_amd/csrc"
mkdir -p _ab/$name
pids=""
for v in "$@"; do
  u=$(basename $v)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $v.hip -o _ab/$name/$u.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/$name/libtpsrhs_$u.so _ab/$name/$u.o ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
echo "built tps_amd/csrc/_ab/$name/: $(ls _ab/$name/*.so | xargs -n1 basename | tr '\n' ' ')"
