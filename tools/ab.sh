#!/bin/bash
# A/B several builds of libtpsrhs.so in one GPU session: tools/ab.sh <workload> lib1.so lib2.so ...
# (each library twice, alternating, so that box-to-box and run-to-run drift shows)
W=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  TPSRHS_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
done
done
