#!/bin/bash
# build a variant of libivx_hip.so with extra -D flags for ivx_join_regions.hip: tools/variant.sh <name> <flags...>
set -eo pipefail
cd "$(dirname "$0")/../datafusion-bio-functions_amd"
N=$1; shift
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../include "$@" -c csrc/ivx_join_regions.hip -o /tmp/jr_$N.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "k_probe_regionsILi1" | grep -E "VGPRs:|ScratchSize|LDS Size" || true
OBJS=$(ls build/*.o | grep -v ivx_join_regions)
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/lib_$N.so $OBJS /tmp/jr_$N.o
echo built lib/lib_$N.so
