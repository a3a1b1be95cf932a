#!/bin/bash
# Developer tool: link a copy of libgtsfm_amd.so with ONE source rebuilt under extra -D flags (kernel variants for A/B runs on the
# GPU box: GTSFM_LIB=tools/bin/<name>.so python ...).   tools/build_variant.sh <name> <source.hip> [-DFLAG=..]...
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p tools/bin
EXTRA=""
[ "$SRC" = "attention_kernels.hip" ] && EXTRA="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function $EXTRA "$@" -c gtsfm_amd/csrc/$SRC -o tools/bin/$NAME.o
OBJS=$(ls gtsfm_amd/csrc/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/bin/$NAME.o -o tools/bin/$NAME.so
rm tools/bin/$NAME.o
echo tools/bin/$NAME.so
