#!/bin/bash
# Two counter passes over every kernel of a workload's deferred-mirror path (profiles/pmc_by_kernel.py sums them).
set -e
WL=${1:-cfg5}
OUT=${2:-gpurun_out/pmc_defer}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --no-cpu --no-frame512 --steps 1 --warmup 0"
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq1" -- python3 $ARGS > "$OUT/sq1.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq3" -- python3 $ARGS > "$OUT/sq3.log" 2>&1
python3 profiles/pmc_by_kernel.py "$OUT"
