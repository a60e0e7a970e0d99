#!/bin/bash
# A/B timing of the cfg5 workload (8192x8192, 4096 spheres, 1 bounce): screen level 4 (table walks only) against 5
# (grid walk for mirror rays), each with and without parked mirror rays.
set -e
OUT=${1:-gpurun_out/ab_cfg5}
mkdir -p $OUT
for screen in 4 5; do
for park in 0; do
  GRAFT_PARK=$park timeout -k 10 200 python bench.py --workload cfg5 --screen $screen --no-cpu --no-frame512 --steps 5 --warmup 2 > $OUT/s${screen}_park$park.json 2> $OUT/s${screen}_park$park.err
  python - $OUT/s${screen}_park$park.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(b["value"]), "Mrays/s", round(b["ms_per_step"],3), "ms", "level", b["config"]["screen_level"])
PY
done
done
