#!/bin/bash
# build a variant of libivx_hip.so with extra -D flags for ONE source file: tools/variant_src.sh <name> <file.hip> <flags...>  -> lib/lib_<name>.so
set -eo pipefail
cd "$(dirname "$0")/../datafusion-bio-functions_amd"
N=$1; F=$2; shift; shift
B=$(basename $F .hip)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../include "$@" -c csrc/$F -o /tmp/${B}_$N.o
OBJS=$(ls build/*.o | grep -v "build/$B.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/lib_$N.so $OBJS /tmp/${B}_$N.o
echo built lib/lib_$N.so
