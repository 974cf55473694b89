#!/bin/bash
# A/B of alternate CORE libraries (the dry-air / table-gas kernels live in libtpsrhs.so): tools/ab_core.sh <workload> base|<dir with libtpsrhs.so> ...
W=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$PWD/$v/libtpsrhs.so
  TPSRHS_LIB=$lib TPSRHS_FAMILY_PATH=$PWD/tps_amd/csrc timeout -k 10 300 python bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
done
done
