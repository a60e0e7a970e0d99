"""Collect the counter CSVs written by profiles/collect_pmc.sh into the record bench.py reads (values of the LAST
trace-kernel launch of each pass):
    python profiles/summarize_pmc.py gpurun_out/pmc_cfg5 cfg5 [screen cap] > profiles/pmc_cfg5.json
The record carries the fingerprint of the kernel sources it was taken on (bench.py prints whether it still matches)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

root, workload = sys.argv[1], sys.argv[2]
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 7
counters, kernel = {}, {}
for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    last = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if "traceKernel" not in row.get("Kernel_Name", ""):
                continue
            last.setdefault(row["Dispatch_Id"], {})[row["Counter_Name"]] = float(row["Counter_Value"])
            kernel = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "Scratch_Size", "VGPR_Count",
                                          "SGPR_Count", "LDS_Block_Size") if k in row}
    if last:
        counters.update(last[max(last, key=int)])
trace, bench = {}, {}
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "traceKernel" in row.get("Name", ""):
                trace = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"])}
for log in glob.glob(os.path.join(root, "trace.log")):
    for line in open(log):
        if line.startswith("{"):
            b = json.loads(line)
            bench = {"rays": b["rays_per_frame"], "tests": b["tests_per_frame"], "ms_per_step": b["ms_per_step"],
                     "frame": b["config"]["frame"], "kernel_source_hash": b.get("kernel_source_hash"),
                     "steps": b["steps"], "warmup": b["warmup"], "kernel_ms": b["roofline"]["kernel_ms"],
                     "screen_level": b["config"]["screen_level"]}
# The trace pass runs the bench with warm-up steps: its timed steps are the LAST launches of the trace kernel, and
# their average is what the bench's own HIP-event time (bench_under_trace.kernel_ms) must agree with; the average of
# the kernel-stats file also holds the counting launch and the warm-up launches.
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    spans = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if "traceKernel" in row.get("Kernel_Name", ""):
                spans.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    spans.sort()
    if spans and trace and bench:
        warm = [d for _, d in spans[-int(bench["steps"]):]]
        trace["warm_calls"] = len(warm)
        trace["warm_average_ns"] = sum(warm) / len(warm)
        trace["warm_min_ns"], trace["warm_max_ns"] = min(warm), max(warm)
v = counters.get
d = {}
if trace:
    d["kernel_ms"] = trace.get("warm_average_ns", trace["average_ns"]) / 1e6
if v("SQ_INSTS_VALU") and bench:
    d["valu_wave_instr_per_wave_ray"] = v("SQ_INSTS_VALU") / (bench["rays"] / 64.0)
if v("SQ_ACTIVE_INST_VALU") and v("SQ_BUSY_CYCLES"):
    # SQ_BUSY_CYCLES counts per shader engine (32 on this part), ACTIVE_INST_VALU per SIMD in units of 4 cycles:
    # busy fraction per SIMD = ACTIVE_INST_VALU * 4 / (BUSY_CYCLES / 32 * 1024)
    d["valu_busy_fraction_per_simd"] = v("SQ_ACTIVE_INST_VALU") * 4 / (v("SQ_BUSY_CYCLES") / 32 * 1024)
if v("SQ_THREAD_CYCLES_VALU") and v("SQ_ACTIVE_INST_VALU"):
    d["lane_utilisation"] = v("SQ_THREAD_CYCLES_VALU") / (64 * v("SQ_ACTIVE_INST_VALU"))
if v("SQ_WAIT_INST_ANY") and v("SQ_WAVE_CYCLES"):
    d["wave_cycles_issue_stalled_fraction"] = v("SQ_WAIT_INST_ANY") / v("SQ_WAVE_CYCLES")
    d["wave_cycles_parked_fraction"] = v("SQ_WAIT_ANY", 0.0) / v("SQ_WAVE_CYCLES")
if v("SQ_LDS_IDX_ACTIVE") and v("SQ_BUSY_CYCLES"):
    # LDS-array cycles summed over the 256 CUs against the kernel's cycles (BUSY_CYCLES / 32 shader engines)
    d["lds_busy_fraction"] = v("SQ_LDS_IDX_ACTIVE") / 256 / (v("SQ_BUSY_CYCLES") / 32)
if v("GRBM_GUI_ACTIVE") and trace:
    # the counter sums the 8 XCDs over the counter pass's own launch, which runs a little slower than the traced ones:
    # an estimate, a few percent low
    d["clock_ghz"] = v("GRBM_GUI_ACTIVE") / 8 / trace.get("warm_average_ns", trace["average_ns"])
if v("WRITE_SIZE") is not None and v("FETCH_SIZE") is not None:
    # both counters are in KiB; FETCH_SIZE tallies 128-byte requests at 64 bytes on this part: doubled
    d["hbm_write_bytes_per_launch"] = v("WRITE_SIZE") * 1024
    d["hbm_fetch_bytes_per_launch"] = 2 * v("FETCH_SIZE") * 1024
    d["hbm_bytes_per_launch"] = d["hbm_write_bytes_per_launch"] + d["hbm_fetch_bytes_per_launch"]
print(json.dumps({"workload": workload, "screen_cap": cap, "frame": bench.get("frame"),
                  "kernel_source_hash": bench.get("kernel_source_hash"), "kernel": kernel, "counters": counters,
                  "kernel_trace": trace, "bench_under_trace": bench, "derived": d,
                  "how": "rocprofv3 --pmc passes of profiles/collect_pmc.sh on `bench.py --workload %s --screen %d "
                         "--steps 1 --warmup 0` (last launch of each pass); kernel trace of the same command with "
                         "--steps %s --warmup %s" % (workload, cap, bench.get("steps"), bench.get("warmup"))}, indent=1))
