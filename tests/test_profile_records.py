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
    kernel_ms = rec["kernel_trace"]["average_ns"] / 1e6
    roof = bench.roofline(rec, "test", kernel_ms, local_tests=1)
    assert 0 < roof["frac"] <= 1
    assert 0 < roof["fp64_issue_frac"] <= 1
    assert roof["frac"] <= 2 * roof["fp64_issue_frac"] + 1e-12  # an instruction is worth at most two flops
    assert 0 < roof["valu_busy"] <= 1 and 0 < roof["lane_utilisation"] <= 1
    # the two independent timings of the record (counter pass, kernel trace) agree
    assert abs(rec["derived"]["kernel_ms"] - kernel_ms) <= 0.25 * kernel_ms
