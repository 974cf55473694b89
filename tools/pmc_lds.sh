#!/bin/bash
# LDS counters of one workload (run on the GPU box): tools/pmc_lds.sh <workload> <outdir-suffix>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-cfg2}
OUT=gpurun_out/pmc_lds_${2:-a}
P="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT" -- $P > /dev/null 2> "$OUT.err"
python3 tools/pmc_summary.py "$OUT"
