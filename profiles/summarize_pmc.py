"""Collect the counter CSVs written by profiles/collect_pmc.sh into one JSON (values of the LAST trace-kernel launch).
    python profiles/summarize_pmc.py gpurun_out/pmc_cfg5 > profiles/<name>_pmc.json"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {}
for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    last = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if "traceKernel" not in row.get("Kernel_Name", ""):
                continue
            key = row["Dispatch_Id"]
            last.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
            out["_kernel"] = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "Scratch_Size") if k in row}
    if last:
        out.update(last[max(last, key=int)])
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "traceKernel" in row.get("Name", ""):
                out["_kernel_trace"] = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"])}
for log in glob.glob(os.path.join(root, "trace.log")):
    for line in open(log):
        if line.startswith("{"):
            b = json.loads(line)
            out["_bench_under_trace"] = {"rays": b["rays_per_frame"], "ms_per_step": b["ms_per_step"]}
v = out.get
d = {}
if v("SQ_INSTS_VALU") and "_bench_under_trace" in out:
    d["valu_wave_instr_per_wave_ray"] = v("SQ_INSTS_VALU") / (out["_bench_under_trace"]["rays"] / 64.0)
if v("SQ_ACTIVE_INST_VALU") and v("SQ_BUSY_CYCLES"):
    # SQ_BUSY_CYCLES counts per shader engine (32 on this part), ACTIVE_INST_VALU per SIMD in units of 4 cycles:
    # busy fraction per SIMD = ACTIVE_INST_VALU * 4 / (BUSY_CYCLES / 32 * 1024)
    d["valu_busy_fraction_per_simd"] = v("SQ_ACTIVE_INST_VALU") * 4 / (v("SQ_BUSY_CYCLES") / 32 * 1024)
if v("SQ_THREAD_CYCLES_VALU") and v("SQ_ACTIVE_INST_VALU"):
    d["lane_utilisation"] = v("SQ_THREAD_CYCLES_VALU") / (64 * v("SQ_ACTIVE_INST_VALU"))
if v("SQ_WAIT_INST_ANY") and v("SQ_WAVE_CYCLES"):
    d["wave_cycles_waiting_fraction"] = v("SQ_WAIT_INST_ANY") / v("SQ_WAVE_CYCLES")
if v("WRITE_SIZE") is not None and v("FETCH_SIZE") is not None:
    d["hbm_write_bytes_per_launch"] = v("WRITE_SIZE") * 1024
    d["hbm_fetch_bytes_per_launch_uncorrected"] = v("FETCH_SIZE") * 1024
out["_derived"] = d
print(json.dumps(out, indent=1))
