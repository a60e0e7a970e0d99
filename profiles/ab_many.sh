#!/bin/bash
# kernel time of one workload for several builds of the library: bash profiles/ab_many.sh <workload> <lib>...
WL=$1; shift
for rep in 1 2; do for lib in "" "$@"; do
  GRAFT_LIB=$lib timeout -k 10 200 python bench.py --workload $WL --no-cpu --no-frame512 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL', '${lib:-in-tree}', round(b['roofline']['kernel_ms'],3), 'ms')"
done; done
