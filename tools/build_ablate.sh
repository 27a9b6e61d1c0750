#!/bin/bash
# Diagnostic builds of the library with parts of the GEMV inner loop removed (see TK_ABLATE in csrc/llm/tk_llm_kernels.hip):
#   build/ablate<n>.so, n = 1 compute only, 2 loads only, 3 neither.  Results are meaningless numerically; timing only.
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
    make -s -C trackiellm_amd/csrc -j8 OBJDIR="$PWD/build/obj_ab$n" OUT="$PWD/build/ablate$n.so" \
        FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$PWD/include -I. -Wall -Wno-unused-function -DTK_ABLATE=$n"
done
