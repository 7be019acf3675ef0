#!/bin/bash
# time several builds of libivx_hip.so inside ONE gpurun call: tools/ab_many.sh "libA.so libB.so ..." [env...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
LIBS=$1; shift
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
  for v in $LIBS; do
    cp $L/$v $L/libivx_hip.so
    echo "== $v (round $round)"
    $R/tools/prof_stats.sh ab_$(basename $v .so)_$round "$@" 2>&1 | grep -E "k_part|k_probe|k_fill"
  done
done
cp $L/.orig.so $L/libivx_hip.so
