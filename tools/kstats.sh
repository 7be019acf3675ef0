#!/bin/bash
# rocprofv3 kernel-trace stats of one python command: tools/kstats.sh <tag> <script> [args..]  -> gpurun_out/ks_<tag>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$R/gpurun_out/ks_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/"$@" > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/ks_${TAG}_kernel_stats.csv && head -25 $f | cut -c1-200
