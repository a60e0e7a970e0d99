#!/bin/bash
# All counter records of a round (run on the GPU box from the repository root): bash profiles/collect_all.sh <tag>
# Writes gpurun_out/pmc_<tag>/<workload>[_level0]/ ; profiles/publish_records.sh turns each into profiles/pmc_*.json
# and copies the kernel-trace summaries beside them.
set -e
TAG=${1:-r3}
bash profiles/collect_pmc.sh cfg3 gpurun_out/pmc_$TAG/cfg3_level0 0
bash profiles/collect_pmc.sh cfg3 gpurun_out/pmc_$TAG/cfg3 7
bash profiles/collect_pmc.sh cfg5 gpurun_out/pmc_$TAG/cfg5 7
bash profiles/collect_pmc.sh cfg4 gpurun_out/pmc_$TAG/cfg4 7
