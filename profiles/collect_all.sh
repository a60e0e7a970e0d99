#!/bin/bash
# All counter records of a round (run on the GPU box from the repository root): bash profiles/collect_all.sh <tag>
# Writes gpurun_out/pmc_<tag>/<workload>[_level0]/ ; profiles/summarize_pmc.py turns each into profiles/pmc_*.json.
set -e
TAG=${1:-r2}
bash profiles/collect_pmc.sh cfg3 gpurun_out/pmc_$TAG/cfg3_level0 0
bash profiles/collect_pmc.sh cfg3 gpurun_out/pmc_$TAG/cfg3 5
bash profiles/collect_pmc.sh cfg5 gpurun_out/pmc_$TAG/cfg5 5
bash profiles/collect_pmc.sh cfg4 gpurun_out/pmc_$TAG/cfg4 5
