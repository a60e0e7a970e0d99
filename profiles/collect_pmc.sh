#!/bin/bash
# Counter passes for the trace kernel of one bench workload (run on the GPU box from the repository root):
#     bash profiles/collect_pmc.sh <workload> <output directory under gpurun_out/>
# Every pass is a run of its own (the profiler does not split counters over passes), kernel trace is not combined
# with counters, and the program itself follows "--".  profiles/summarize_pmc.py turns the CSVs into one JSON.
set -e
WL=${1:-cfg4}
OUT=${2:-gpurun_out/pmc_$WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --no-cpu --no-frame512 --steps 1 --warmup 0"
pass() {
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1
    echo "pass $name done"
}
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32
pass sq3 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE GRBM_COUNT
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1
echo "trace done"
