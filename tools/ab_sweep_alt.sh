for rep in 1 2; do
for alt in 0 1; do
for W in argon_p3 cfg2; do
  fp=$PWD/tps_amd/csrc/_ab/alt
  TPSRHS_SWEEP_ALT=$alt TPSRHS_FAMILY_PATH=$fp timeout -k 10 300 python bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W alt=$alt', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()}, 'rk4', d.get('time_loop',{}).get('ms_per_rk4_step'))"
done
done
done
