#!/bin/bash
# After profiles/collect_all.sh <tag> has run on a GPU box and gpurun_out/pmc_<tag>/ has come back:
#     bash profiles/publish_records.sh <tag>
# writes profiles/pmc_<workload>[_level0].json (what bench.py reads), profiles/<tag>_<workload>_kernel_stats.csv (the
# rocprofv3 --kernel-trace --stats summary of the same command) and profiles/<tag>_<workload>_trace_kernel_rows.csv
# (start and duration of every launch of the trace kernel in that run: the last `steps` rows are the timed steps).
set -e
TAG=${1:-r3}
for wl in cfg3_level0:cfg3:0 cfg3:cfg3:7 cfg5:cfg5:7 cfg4:cfg4:7; do
  dir=${wl%%:*}; rest=${wl#*:}; name=${rest%%:*}; cap=${rest#*:}
  src=gpurun_out/pmc_$TAG/$dir
  [ -d "$src" ] || continue
  # (gpurun merges what a run wrote into what is already here: a second collection under the same tag leaves two sets)
  if [ "$(ls $src/trace/*/*kernel_stats.csv | wc -l)" != 1 ]; then echo "$src holds more than one collection: remove it and collect again, or use another tag" >&2; exit 1; fi
  python profiles/summarize_pmc.py $src $name $cap > profiles/pmc_$dir.json
  cp $src/trace/*/*kernel_stats.csv profiles/${TAG}_${dir}_kernel_stats.csv
  python - "$src" > profiles/${TAG}_${dir}_trace_kernel_rows.csv <<'PY'
import csv, glob, os, sys
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "traceKernel" in row["Kernel_Name"]:
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row["Kernel_Name"],
                         row["Scratch_Size"], row["VGPR_Count"], row["Workgroup_Size_X"], row["Grid_Size_X"]))
print("start_ns,duration_ns,kernel,scratch_bytes_per_lane,vgprs,workgroup,grid")
for r in sorted(rows):
    print("%d,%d,\"%s\",%s,%s,%s,%s" % r)
PY
  grep "^{" $src/trace.log | tail -n 1 > profiles/${TAG}_bench_under_trace_$dir.json
done
