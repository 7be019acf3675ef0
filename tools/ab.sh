#!/bin/bash
# A/B two builds of libivx_hip.so inside ONE gpurun call (boxes differ by >10 %): tools/ab.sh libA.so libB.so [env...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
A=$1; B=$2; shift 2
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
  for v in $A $B; do
    cp $L/$v $L/libivx_hip.so
    echo "== $v (round $round)"
    tools/prof_stats.sh ab_$(basename $v .so)_$round "$@" 2>&1 | grep -E "k_part|k_probe"
  done
done
cp $L/.orig.so $L/libivx_hip.so
