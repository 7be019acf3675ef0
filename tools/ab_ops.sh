#!/bin/bash
# A/B library builds on count/coverage and the fill probe inside one session: tools/ab_ops.sh "libs"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
for v in $1; do
  cp $L/$v $L/libivx_hip.so
  echo "== $v (round $round)"
  OPS=count,coverage python3 tools/ops_perf.py 2>&1 | grep probe
  tools/prof_stats.sh abo_$(basename $v .so)_$round MODE=fill REPS=3 2>&1 | grep -E "k_probe_regions<1>"
done
done
cp $L/.orig.so $L/libivx_hip.so
