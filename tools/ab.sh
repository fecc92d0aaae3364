#!/bin/bash
# A/B of library builds on ONE box (boxes differ by up to 10 %): tools/ab.sh <workload> <reps> libA.so libB.so ...
# ("default" = the in-tree library).  Prints windows/s, conv TFLOP/s and per-class kernel ms for every run.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=$1; REPS=$2; shift 2
cd "$R"
for i in $(seq 1 $REPS); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset CAMAL_HIP_LIB; else export CAMAL_HIP_LIB=$lib; fi
    timeout -k 10 150 python bench.py --workload $WL --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
    python - "$lib" <<PY
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:34s} {d['value']:10.0f} w/s  conv {d['roofline']['achieved']:7.1f}  post {d['roofline_post']['achieved']:6.0f} GB/s ", {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()}, flush=True)
PY
  done
done
