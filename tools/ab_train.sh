#!/bin/bash
# A/B of library builds on ONE box for the training step: tools/ab_train.sh <reps> libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=$1; shift
cd "$R"
for i in $(seq 1 $REPS); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset CAMAL_HIP_LIB; else export CAMAL_HIP_LIB=$lib; fi
    timeout -k 10 150 python bench.py --workload train_fp32 --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
    python - "$lib" <<PY
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:34s} {d['value']:8.0f} w/s  {d['ms_per_step']:7.2f} ms/step  frac {d['roofline']['frac']:.3f}", flush=True)
PY
  done
done
