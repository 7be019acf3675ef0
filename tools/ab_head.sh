#!/bin/bash
# headline probe, libs alternated: tools/ab_head.sh "libA.so libB.so"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2 3; do
  for v in $1; do
    cp $L/$v $L/libivx_hip.so
    echo "round $round $v fill: $(MODE=fill REPS=5 python3 tools/probe_only.py 2>&1 | grep kernel_ms | tail -1 | sed 's/.*pairs/pairs/')"
  done
done
cp $L/.orig.so $L/libivx_hip.so
