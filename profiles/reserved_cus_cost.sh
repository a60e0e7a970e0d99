#!/bin/bash
# What leaving r compute units free costs a one-GPU trace: bash profiles/reserved_cus_cost.sh [workloads...]
# Prints kernel milliseconds for r = 0, 2, 4, 8, twice each, alternating.
WLS=${@:-"cfg4"}
for wl in $WLS; do
  for rep in 1 2; do
    for r in 0 2 4 8; do
      timeout -k 10 200 python bench.py --workload $wl --reserve-cus $r --no-cpu --no-frame512 --no-host-frame --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl reserved', $r, round(b['roofline']['kernel_ms'],4), 'ms')" || exit 1
    done
  done
done
