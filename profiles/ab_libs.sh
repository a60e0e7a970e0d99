#!/bin/bash
# A/B timing of two builds of the library on one box: bash profiles/ab_libs.sh <other .so> [workloads...]
# Prints kernel milliseconds per workload for the in-tree library (screen cap $CAP, default 7) and for the other one
# (screen cap $OTHER_CAP, default $CAP), alternating, $REPS times each (default 2).
OTHER=$1; shift
WLS=${@:-"cfg3 cfg4 cfg5"}
CAP=${CAP:-7}
OTHER_CAP=${OTHER_CAP:-$CAP}
for wl in $WLS; do
  for rep in $(seq 1 ${REPS:-2}); do
    for lib in "" "$OTHER"; do
      cap=$CAP; [ -n "$lib" ] && cap=$OTHER_CAP
      GRAFT_LIB=$lib timeout -k 10 200 python bench.py --workload $wl --screen $cap --no-cpu --no-frame512 --no-host-frame --steps ${STEPS:-5} --warmup 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl cap $cap', '${lib:-in-tree}', round(b['roofline']['kernel_ms'],4), 'ms', round(b['value']), 'Mrays/s', 'level', b['config']['screen_level'])" || exit 1
    done
  done
done
