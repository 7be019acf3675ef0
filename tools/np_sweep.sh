#!/bin/bash
# per-row kernel cost vs probe rows (does a chunk that fits the Infinity Cache run faster?)
for np in $1; do
  echo "== NP=$np"
  tools/prof_stats.sh np_$np NP=$np MODE=fill REPS=5 IVX_JOIN_PATH=regions 2>&1 | grep -E "k_part|k_probe"
done
