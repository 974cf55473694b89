# A/B of the fused trace epilogue in the time loop: bench's time_loop (hipGraph replay of tpsrhs_advance) with and without
for rep in 1 2; do
for f in 0 1; do
for W in argon_p3 cfg2; do
  TPSRHS_FUSE_TRACES=$f timeout -k 10 300 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W fuse=$f', 'Mult', round(d['ms_per_step'],4), 'rk4 step', round(d['time_loop']['ms_per_rk4_step'],4), 'nan', d['time_loop']['nan_entries'])"
done
done
done
