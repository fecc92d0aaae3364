#!/bin/bash
# Kernel-trace profile of one bench.py workload on the GPU box (run through gpurun):
#   tools/profile.sh <workload> <tag> [extra bench.py args]
# leaves gpurun_out/prof_<tag>/*kernel_stats.csv and gpurun_out/prof_<tag>.log
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$TAG" -o "$TAG" -- \
    python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --workload "$WL" "$@" > "$R/gpurun_out/prof_$TAG.log" 2>&1
rc=$?
grep '^{' "$R/gpurun_out/prof_$TAG.log" | cut -c1-600
exit $rc
