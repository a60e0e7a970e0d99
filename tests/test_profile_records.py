"""The counter records bench.py reads (profiles/pmc_*.json) belong to the kernel sources in the tree, and the
executed-work fractions derived from them are fractions: at most 1, whatever the instruction mix."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_*.json")))


def test_there_are_records_for_the_bench_workloads():
    names = {os.path.basename(p) for p in RECORDS}
    assert {"pmc_cfg3.json", "pmc_cfg3_level0.json", "pmc_cfg4.json", "pmc_cfg5.json"} <= names


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_record_matches_the_sources_and_gives_fractions(path):
    rec = json.load(open(path))
    assert rec["kernel_source_hash"] == bench.kernel_source_hash(), \
        "%s was recorded on other kernel sources: run profiles/collect_all.sh and profiles/summarize_pmc.py" % path
    kernel_ms = rec["kernel_trace"]["warm_average_ns"] / 1e6
    roof = bench.roofline(rec, "test", kernel_ms, local_tests=1)
    assert 0 < roof["frac"] <= 1
    assert 0 < roof["fp64_issue_frac"] <= 1
    assert roof["frac"] <= 2 * roof["fp64_issue_frac"] + 1e-12  # an instruction is worth at most two flops
    # (the busy fraction divides two counters of different granularity -- busy cycles per shader engine, VALU cycles per
    # SIMD -- and can come out a few percent above 1 on a launch whose engines finish at different times)
    assert 0 < roof["valu_busy"] <= 1.1 and 0 < roof["lane_utilisation"] <= 1
    # The trace pass runs the bench with warm-up steps; its timed steps are the last launches in the kernel trace.  Their
    # average and the bench's own HIP-event time for the same steps agree to 3 % (or 15 microseconds on launches of a
    # fraction of a millisecond: the events also bracket the memset of the ticket counters and the gap before the launch).
    trace, under = rec["kernel_trace"], rec["bench_under_trace"]
    assert trace["warm_calls"] == under["steps"] >= 10
    warm_ms = trace["warm_average_ns"] / 1e6
    assert abs(warm_ms - under["kernel_ms"]) <= max(0.03 * warm_ms, 0.015), (warm_ms, under["kernel_ms"])
    assert abs(rec["derived"]["kernel_ms"] - warm_ms) < 1e-9


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_kernel_stats_file_beside_the_record(path):
    """The rocprofv3 --kernel-trace --stats summary of the record's trace pass is kept beside it, and the rows of the
    trace kernel extracted from the trace give the record's warm average."""
    import csv
    import glob as g
    stem = os.path.basename(path)[len("pmc_"):-len(".json")]
    stats = g.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.csv" % stem))
    rows = g.glob(os.path.join(ROOT, "profiles", "r*_%s_trace_kernel_rows.csv" % stem))
    assert stats and rows, "no kernel-trace summary for %s: run profiles/publish_records.sh" % stem
    rec = json.load(open(path))
    durations = [int(r["duration_ns"]) for r in csv.DictReader(open(sorted(rows)[-1]))]
    steps = rec["bench_under_trace"]["steps"]
    assert abs(sum(durations[-steps:]) / steps - rec["kernel_trace"]["warm_average_ns"]) < 1.0
    with open(sorted(stats)[-1]) as f:
        trace_rows = [r for r in csv.DictReader(f) if "traceKernel" in r["Name"]]
    # (a job may run two kernels: the bench's counting launch asks for ray counts, its timed launches for the frame alone)
    assert trace_rows and sum(int(r["Calls"]) for r in trace_rows) == len(durations)
