#!/bin/bash
# Run on the GPU box (gpurun): kernel trace + the PMC passes of one bench workload, each in its own
# rocprofv3 run (counters never share a run with the trace).  Usage: tools/profile_round.sh <tag> <workload>
set -e
TAG=${1:-r01}
W=${2:-argon_p3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${W}
mkdir -p "$OUT"
B="python3 bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline --no-other-workloads"  # the default K and W of bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $B > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
echo "trace done"
P="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $P > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $P > /dev/null 2> "$OUT/pmc_write.err"
echo "hbm counters done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d "$OUT/pmc_sq1" -- $P > /dev/null 2> "$OUT/pmc_sq1.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $P > /dev/null 2> "$OUT/pmc_sq2.err"
echo "sq counters done"
$B > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
python3 tools/profile_summary.py "$OUT" "$TAG" "$W"
# what the summary wrote under profiles/ lives on the GPU box only: hand it back through gpurun_out/
mkdir -p gpurun_out/profiles_back
cp profiles/${TAG}_${W}_* profiles/hbm_traffic.json gpurun_out/profiles_back/ 2>/dev/null || true
