#!/bin/bash
# Everything the round's profiles/ directory is made of, in one GPU call:  tools/round_profiles.sh <tag>
# kernel-trace stats + bench lines for the workloads, FETCH/WRITE passes for the traffic figures.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
cd "$R"
set -e
for wl in resnet_fp32 ensemble_bf16 long_bf16 train_fp32; do
  bash tools/profile.sh $wl ${TAG}_$wl > /dev/null
done
for wl in resnet_fp32 ensemble_bf16; do
  bash tools/pmc.sh $wl ${TAG}_${wl}_fetch "FETCH_SIZE" "conv_kernel|block|post_kernel"
  bash tools/pmc.sh $wl ${TAG}_${wl}_write "WRITE_SIZE" "conv_kernel|block|post_kernel"
done
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench_resnet_fp32.json
for wl in ensemble_bf16 ensemble_fp32 long_bf16 train_fp32; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/${TAG}_bench_$wl.json
done
echo profiles done
