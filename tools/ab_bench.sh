#!/bin/bash
# A/B bench.py between library builds inside one session: tools/ab_bench.sh "lib1.so lib2.so"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
cp $L/libivx_hip.so $L/.orig.so
for round in 1 2; do
for v in $1; do
  cp $L/$v $L/libivx_hip.so
  echo "== $v (round $round)"
  python3 bench.py --cpu-sample 1000000 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'build_ms', round(d['roofline']['build_ms'],4))"
done
done
cp $L/.orig.so $L/libivx_hip.so
