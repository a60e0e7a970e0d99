#!/bin/bash
# Kernel time of workloads for several builds of the library on one box:
#     WLS="cfg4 cfg3" bash profiles/ab_many.sh <lib>...       ("" or "in-tree" = the library in the tree)
# Each build is timed $REPS times (default 2), alternating; the parity tests are not run here.
WLS=${WLS:-cfg4}
for wl in $WLS; do for rep in $(seq 1 ${REPS:-2}); do for lib in "$@"; do
  [ "$lib" = in-tree ] && lib=""
  GRAFT_LIB=$lib timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-frame512 --no-host-frame --steps ${STEPS:-5} --warmup 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '${lib:-in-tree}', round(b['roofline']['kernel_ms'],4), 'ms', 'level', b['config']['screen_level'])" || exit 1
done; done; done
