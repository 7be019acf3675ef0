#!/bin/bash
# ablation of the fill probe inside ONE session (library built with -DIVX_ABLATE, e.g. tools/variant.sh abl -DIVX_ABLATE,
# then copied over lib/libivx_hip.so): tools/ablate.sh "0 16 4 20" [env...]
BITS=$1; shift
for b in $BITS; do
  echo "== IVX_DBG=$b"
  tools/prof_stats.sh abl_$b IVX_DBG=$b "$@" 2>&1 | grep -E "k_part|k_probe"
done
