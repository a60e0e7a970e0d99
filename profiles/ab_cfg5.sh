#!/bin/bash
# A/B timing of the cfg5 workload (8192x8192, 4096 spheres, 1 bounce) with and without parked mirror rays.
set -e
OUT=${1:-gpurun_out/ab_cfg5}
mkdir -p $OUT
for park in 0 1; do
  GRAFT_PARK=$park timeout -k 10 200 python bench.py --workload cfg5 --no-cpu --no-frame512 --steps 5 --warmup 2 > $OUT/park$park.json 2> $OUT/park$park.err
  python - $OUT/park$park.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(b["value"]), "Mrays/s", round(b["ms_per_step"],3), "ms")
PY
done
