#!/bin/bash
# A/B library builds across build-side shapes inside one session: tools/ab_dense.sh "libs"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
cp $L/libivx_hip.so $L/.orig.so
for v in $1; do
  cp $L/$v $L/libivx_hip.so
  echo "== $v"
  for cfg in "100000000 1000000 1000" "100000000 1000000 5000" "100000000 1000000 20000" "100000000 10000000 1000" "50000000 50000000 1000"; do
    set -- $cfg
    NP=$1 NB=$2 BMEAN=$3 MODE=fill REPS=2 python3 tools/probe_only.py 2>&1 | grep -v amdgpu | tail -1 | awk -v c="$cfg" '{print c, $0}'
  done
done
cp $L/.orig.so $L/libivx_hip.so
