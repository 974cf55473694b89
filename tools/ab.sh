#!/bin/bash
# A/B several builds of a kernel family in one GPU session: tools/ab.sh <workload> <variant dir | base> ...
#   tools/ab.sh argon_p3 base tps_amd/csrc/_ab/w3       (a variant = a directory of tools/build_variant.sh; base = the regular build)
# (each build twice, alternating, so that box-to-box and run-to-run drift shows)
W=$1; shift
for rep in 1 2; do
for v in "$@"; do
  fp=""; [ "$v" != base ] && fp=$PWD/$v
  TPSRHS_FAMILY_PATH=$fp timeout -k 10 300 python bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
done
done
