#!/bin/bash
# Build a variant of the library for an A/B run ONLY if the ring checker passes on its device assembly:
#   tools/safe_build.sh build/lib_variant.so [-DKNOB=1 ...]
# (a variant whose in-flight weight-ring registers the compiler copies or spills faults the GPU: check on the CPU first)
set -e -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
ASM=$(mktemp /tmp/camal_XXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o "$ASM" "$R/camal_amd/csrc/camal_hip.hip" 2>/dev/null
if ! python3 "$R/tools/check_ring.py" "$ASM" | tail -3; then
  echo "ring checker: findings -- not building $OUT"; rm -f "$ASM"; exit 1
fi
rm -f "$ASM"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o "$OUT" "$R/camal_amd/csrc/camal_hip.hip"
echo "built $OUT"
