#!/bin/bash
# time several builds of libivx_hip.so on tools/ops_perf.py inside ONE gpurun call: tools/ab_many_ops.sh "libA.so libB.so" OPS=count [env...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
LIBS=$1; shift
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
  for v in $LIBS; do
    cp $L/$v $L/libivx_hip.so
    echo "== $v (round $round)"
    $R/tools/prof_ops.sh ab_$(basename $v .so)_$round "$@" 2>&1 | grep -E "k_rv_fast|k_probe_regions|k_unpermute"
  done
done
cp $L/.orig.so $L/libivx_hip.so
