#!/bin/bash
# Baseline on the GPU box: quick parity check, bench lines, level-0 (exact double loop) profile of cfg3,
# LDS counters of cfg4.  Every GPU step has its own time limit; the script stops at the first failure.
set -e
OUT=gpurun_out/r2_base
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "depth or default_scene" > $OUT/pytest.log 2>&1
echo "pytest ok"
for wl in cfg3 cfg4 cfg5; do
  timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-frame512 --steps 5 --warmup 2 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  echo "bench $wl ok"
done
export GRAFT_SCREEN=0
timeout -k 10 200 python bench.py --workload cfg3 --no-cpu --no-frame512 --steps 5 --warmup 2 > $OUT/bench_cfg3_level0.json 2> $OUT/bench_cfg3_level0.err
echo "bench level0 ok"
ARGS="bench.py --workload cfg3 --no-cpu --no-frame512 --steps 2 --warmup 1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l0_trace -- python3 $ARGS > $OUT/l0_trace.log 2>&1
echo "level0 trace ok"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/l0_sq1 -- python3 $ARGS > $OUT/l0_sq1.log 2>&1
echo "level0 sq1 ok"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/l0_sq2 -- python3 $ARGS > $OUT/l0_sq2.log 2>&1
echo "level0 sq2 ok"
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/l0_sq3 -- python3 $ARGS > $OUT/l0_sq3.log 2>&1
echo "level0 sq3 ok"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/l0_grbm -- python3 $ARGS > $OUT/l0_grbm.log 2>&1
echo "level0 grbm ok"
unset GRAFT_SCREEN
ARGS="bench.py --workload cfg4 --no-cpu --no-frame512 --steps 1 --warmup 0"
timeout -k 10 240 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/cfg4_lds -- python3 $ARGS > $OUT/cfg4_lds.log 2>&1
echo "cfg4 lds ok"
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
echo "all done"
