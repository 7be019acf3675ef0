#!/bin/bash
# A/B library builds over several build-side shapes (kernel_ms of count and fill): tools/ab_layout.sh "libA.so libB.so" [configs...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; L=$R/datafusion-bio-functions_amd/lib
LIBS=$1; shift
CFGS=("$@")
if [ ${#CFGS[@]} -eq 0 ]; then CFGS=("NP=100000000 NB=1000000 BMEAN=1000" "NP=100000000 NB=10000000 BMEAN=1000" "NP=50000000 NB=50000000 BMEAN=1000" "NP=100000000 NB=1000000 BMEAN=20000" "NP=100000000 NB=200000 BMEAN=200000" "NP=20000000 NB=5000000 BMEAN=300"); fi
cp $L/libivx_hip.so $L/.orig.so
for cfg in "${CFGS[@]}"; do
  for v in $LIBS; do
    cp $L/$v $L/libivx_hip.so
    for mode in count fill; do
      echo "$cfg $v $mode: $(env $cfg MODE=$mode REPS=3 python3 tools/probe_only.py 2>&1 | grep kernel_ms | tail -1 | sed 's/.*pairs/pairs/')"
    done
  done
done
cp $L/.orig.so $L/libivx_hip.so
