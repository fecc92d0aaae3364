#!/bin/bash
# One rocprofv3 counter pass over a bench.py workload (run through gpurun):
#   tools/pmc.sh <workload> <tag> "<counters>" [kernel regex]
# leaves gpurun_out/pmc_<tag>/*counter_collection.csv
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=$1; TAG=$2; CTR=$3; RE=${4:-.}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $CTR --kernel-include-regex "$RE" --output-format csv -d "$R/gpurun_out/pmc_$TAG" -o "$TAG" -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload "$WL" > "$R/gpurun_out/pmc_$TAG.log" 2>&1
