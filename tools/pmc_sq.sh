#!/bin/bash
# SQ counters of one workload (run on the GPU box): tools/pmc_sq.sh <workload> <outdir-suffix>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-argon_p3}
OUT=gpurun_out/pmc_sq_${2:-a}
P="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/p1" -- $P > /dev/null 2> "$OUT.1.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d "$OUT/p2" -- $P > /dev/null 2> "$OUT.2.err"
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_FLAT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p3" -- $P > /dev/null 2> "$OUT.3.err"
python3 tools/pmc_summary.py "$OUT"
