#!/bin/bash
# Final measurements on the GPU box: bench lines per workload (counter records are collected beforehand, see below).
set -e
# (counter records first: bash profiles/collect_all.sh <tag>, then profiles/summarize_pmc.py per workload, so that the bench lines find records of this build)
mkdir -p gpurun_out/r2_final
timeout -k 10 600 python bench.py > gpurun_out/r2_final/bench_cfg4.json 2> gpurun_out/r2_final/bench_cfg4.err
for wl in cfg3 cfg5; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-frame512 > gpurun_out/r2_final/bench_$wl.json 2> gpurun_out/r2_final/bench_$wl.err
done
timeout -k 10 300 python bench.py --workload cfg3 --screen 0 --steps 10 --warmup 2 --no-frame512 --no-cpu > gpurun_out/r2_final/bench_cfg3_level0.json 2> gpurun_out/r2_final/bench_cfg3_level0.err
timeout -k 10 300 python bench.py --workload default --steps 20 --warmup 2 --no-frame512 > gpurun_out/r2_final/bench_default.json 2> gpurun_out/r2_final/bench_default.err
echo "final done"
