#!/bin/bash
# diagnostic builds of the LLM kernel file: tools/build_variant.sh NAME "-DTK_...=..." ... -> build/variants/libtrackie_NAME.so
# (select with TK_MI355X_LIB=<path>; the product build is `make -C trackiellm_amd/csrc`, which must be up to date first).
# The product source carries NO diagnostic branches: tools/diag/g32_diagnostics.patch adds them (-DTK_G32_ABL=<bits> timing-only ablations,
# -DTK_G32_CLOCK=1|2 in-kernel stamps + the tk_debug_g32_* exports read by tools/time_gemv.py --stamps, -DTK_G32_PK, -DTK_G32_BALANCE=0,
# -DTK_G32_WIDE_STORE=0, -DTK_G32_NT_STORE=1, -DTK_ATT_ABL) to a scratch copy that only this script compiles.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants; mkdir -p $OUT/src/llm $OUT/src/common
cp $ROOT/trackiellm_amd/csrc/llm/*.h $OUT/src/llm/
cp $ROOT/trackiellm_amd/csrc/common/*.h $OUT/src/common/
cp $ROOT/trackiellm_amd/csrc/llm/tk_llm_kernels.hip $OUT/src/llm/tk_llm_kernels.hip
(cd $OUT/src && patch -s -p0 llm/tk_llm_kernels.hip < $ROOT/tools/diag/g32_diagnostics.patch)
cd $ROOT/trackiellm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -I$ROOT/include -I$OUT/src/llm -I. -Wall -Wno-unused-function "$@" -c $OUT/src/llm/tk_llm_kernels.hip -o $OUT/k_$NAME.o
OBJS=$(find $ROOT/build/obj -name '*.o' | grep -v 'llm/tk_llm_kernels.hip.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libtrackie_$NAME.so $OBJS $OUT/k_$NAME.o -lpthread
echo built $OUT/libtrackie_$NAME.so
