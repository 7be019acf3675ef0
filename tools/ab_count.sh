#!/bin/bash
# count-probe skeleton vs batch size inside one session: tools/ab_count.sh "libs" "dbg bits"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
cp $L/libivx_hip.so $L/.orig.so
for v in $1; do
  cp $L/$v $L/libivx_hip.so
  for b in $2; do
    echo "== $v IVX_DBG=$b"
    tools/prof_stats.sh abc_$(basename $v .so)_$b IVX_DBG=$b MODE=count REPS=3 2>&1 | grep -E "k_probe"
  done
done
cp $L/.orig.so $L/libivx_hip.so
