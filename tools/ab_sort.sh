#!/bin/bash
# kernel times of several builds of libivx_hip.so on tools/ops_perf.py inside ONE gpurun call: tools/ab_sort.sh "libA.so libB.so" "<grep pattern>" OPS=merge [env...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
LIBS=$1; PAT=$2; shift; shift
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
  for v in $LIBS; do
    cp $L/$v $L/libivx_hip.so
    echo "== $v (round $round)"
    $R/tools/prof_ops.sh ab_$(basename $v .so)_$round "$@" 2>&1 | grep -E "$PAT"
    grep -E "^merge|^cluster|^subtract|^nearest" $R/gpurun_out/po_ab_$(basename $v .so)_$round.log
  done
done
cp $L/.orig.so $L/libivx_hip.so
