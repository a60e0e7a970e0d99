#!/bin/bash
# Counter passes for the trace kernel of one bench workload (run on the GPU box from the repository root):
#     bash profiles/collect_pmc.sh <workload> <output directory under gpurun_out/> [screen cap, default 7]
# Every pass is a run of its own (the profiler does not split counters over passes), kernel trace is not combined
# with counters, and the program itself follows "--".  A pass that exceeds its limit gets TERM first, so that the
# profiler can take its child down with it, and KILL ten seconds later; the script stops at the first failed pass.
# profiles/summarize_pmc.py turns the CSVs into profiles/pmc_<workload>[_level<cap>].json, which bench.py reads.
set -e
WL=${1:-cfg4}
OUT=${2:-gpurun_out/pmc_$WL}
CAP=${3:-7}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --screen $CAP --no-cpu --no-frame512 --no-host-frame --steps 1 --warmup 0"
pass() {
    local name=$1; shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1
    echo "pass $name done"
}
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass sq2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32
pass sq3 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass sq4 SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE GRBM_COUNT
# the kernel trace runs the bench with warm-up steps, so that the traced launches are what the bench itself times
TRACE_ARGS="bench.py --workload $WL --screen $CAP --no-cpu --no-frame512 --no-host-frame --steps 20 --warmup 5"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $TRACE_ARGS > "$OUT/trace.log" 2>&1
echo "trace done"
