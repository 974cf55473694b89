#!/bin/bash
# Probe build: the reduced translation unit tps_amd/csrc/_probe/plasma_3d_n7.hip with extra flags, linked with the
# regular objects in place of plasma_3d_n7.o.   tools/probe_build.sh <name> "<flags>"   -> tps_amd/csrc/_ab/<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../tps_amd/csrc"
mkdir -p _ab
objs=""
for o in _obj/*.o; do [ "$(basename $o)" = "plasma_3d_n7.o" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c _probe/plasma_3d_n7.hip -o _ab/probe_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/$name.so $objs _ab/probe_$name.o
rm -f _ab/probe_$name.o
echo "built tps_amd/csrc/_ab/$name.so"
