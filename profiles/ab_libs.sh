#!/bin/bash
# A/B timing of two builds of the library on one box: bash profiles/ab_libs.sh <other .so> [workloads...]
# Prints kernel milliseconds per workload and screen cap for the in-tree library and for the other one, twice each.
OTHER=$1; shift
WLS=${@:-"cfg3 cfg4 cfg5"}
for wl in $WLS; do
  for rep in 1 2; do
    for lib in "" "$OTHER"; do
      for cap in 5 0; do
        if [ "$cap" = 0 ] && [ "$wl" != cfg3 ]; then continue; fi
        GRAFT_LIB=$lib timeout -k 10 200 python bench.py --workload $wl --screen $cap --no-cpu --no-frame512 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl cap $cap', '${lib:-in-tree}', round(b['roofline']['kernel_ms'],4), 'ms', round(b['value']), 'Mrays/s')" || exit 1
      done
    done
  done
done
