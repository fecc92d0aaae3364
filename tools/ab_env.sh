#!/bin/bash
# A/B of environment settings on ONE box: tools/ab_env.sh <workload> <reps> "VAR=a" "VAR=b" ... ("-" = nothing set)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=$1; REPS=$2; shift 2
cd "$R"
for i in $(seq 1 $REPS); do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi
    env $pre timeout -k 10 150 python bench.py --workload $WL --no-cpu-baseline --no-extras > gpurun_out/abe_tmp.json 2> gpurun_out/abe_tmp.err || { tail -5 gpurun_out/abe_tmp.err; exit 1; }
    python - "$kv" <<PY
import json, sys
d = json.load(open("gpurun_out/abe_tmp.json"))
print(f"{sys.argv[1]:40s} {d['value']:10.0f} w/s  {d['ms_per_step']:8.3f} ms/step", flush=True)
PY
  done
done
