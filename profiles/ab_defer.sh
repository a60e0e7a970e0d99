#!/bin/bash
# A/B timing of a workload with mirror rays traced in launches of their own (GRAFT_DEFER=1) against one launch (0),
# then the per-kernel times of the deferred path.
#   profiles/ab_defer.sh [workload] [outdir]
set -e
WL=${1:-cfg5}
OUT=${2:-gpurun_out/ab_defer}
mkdir -p $OUT
for defer in 0 1; do
  GRAFT_DEFER=$defer timeout -k 10 200 python bench.py --workload $WL --no-cpu --no-frame512 --steps 5 --warmup 2 > $OUT/${WL}_defer$defer.json 2> $OUT/${WL}_defer$defer.err
  python - $OUT/${WL}_defer$defer.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(b["value"]), "Mrays/s", round(b["ms_per_step"],3), "ms", "level", b["config"]["screen_level"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o $WL -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --no-cpu --no-frame512 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/${WL}_trace.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
