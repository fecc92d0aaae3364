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
  bash tools/pmc.sh $wl ${TAG}_${wl}_fetch "FETCH_SIZE" "conv_kernel|block|post_"
  bash tools/pmc.sh $wl ${TAG}_${wl}_write "WRITE_SIZE" "conv_kernel|block|post_"
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_all.json
echo profiles done
