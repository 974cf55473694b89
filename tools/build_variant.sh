#!/bin/bash
# Build a variant of libtpsrhs.so for A/B timing: the listed translation units are recompiled with extra
# flags, every other object comes from the regular build.
#   tools/build_variant.sh <name> "<flags>" unit1 [unit2 ...]     -> tps_amd/csrc/_ab/<name>.so
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../tps_amd/csrc"
mkdir -p _ab/$name
objs=""
for o in _obj/*.o; do
  u=$(basename $o .o)
  skip=0
  for v in "$@"; do [ "$v" = "$u" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
pids=""
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $v.hip -o _ab/$name/$v.o &
  pids="$pids $!"
  objs="$objs _ab/$name/$v.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/$name.so $objs
echo "built tps_amd/csrc/_ab/$name.so"
