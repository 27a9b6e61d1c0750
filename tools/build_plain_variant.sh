#!/bin/bash
# A/B builds of the LLM kernel file WITHOUT the diagnostics patch: tools/build_plain_variant.sh NAME "-DTK_...=..." ... -> build/variants/libtrackie_NAME.so
# (select with TK_MI355X_LIB=<path>; `make -C trackiellm_amd/csrc` must be up to date first: every other object is taken from build/obj)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants; mkdir -p $OUT
cd $ROOT/trackiellm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -I$ROOT/include -I. -Wall -Wno-unused-function "$@" -c ${SRC:-llm/tk_llm_kernels.hip} -o $OUT/k_$NAME.o
OBJS=$(find $ROOT/build/obj -name '*.o' | grep -v 'llm/tk_llm_kernels.hip.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtrackie_$NAME.so $OBJS $OUT/k_$NAME.o -lpthread -ldl
echo built $OUT/libtrackie_$NAME.so
