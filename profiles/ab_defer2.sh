#!/bin/bash
# kernel times of the deferred-mirror path for several builds of the library:  profiles/ab_defer2.sh <workload> <lib>...
set -e
WL=$1; shift
OUT=gpurun_out/ab_defer2
mkdir -p $OUT
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  GRAFT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $WL --no-cpu --no-frame512 --steps 5 --warmup 2 > $OUT/${WL}_$tag.json 2> $OUT/${WL}_$tag.err
  python - $OUT/${WL}_$tag.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(b["value"]), "Mrays/s", round(b["ms_per_step"],3), "ms", "level", b["config"]["screen_level"])
PY
  GRAFT_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o $WL -- python3 bench.py --workload $WL --no-cpu --no-frame512 --steps 3 --warmup 1 > $OUT/${WL}_${tag}_trace.log 2>&1
  find $OUT/trace_$tag -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1,2,4 | grep -i "graft" | cut -c1-150
done
