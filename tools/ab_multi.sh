#!/bin/bash
# A/B several library builds in one session: tools/ab_multi.sh "lib1.so lib2.so ..." env...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
LIBS=$1; shift
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
  for v in $LIBS; do
    cp $L/$v $L/libivx_hip.so
    echo "== $v (round $round)"
    tools/prof_stats.sh abm_$(basename $v .so)_$round "$@" 2>&1 | grep -E "k_part|k_probe"
  done
done
cp $L/.orig.so $L/libivx_hip.so
