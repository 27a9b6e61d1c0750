#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the CPU test-suite (GPU sanitizers are not available on this pool): builds the library with
# -fsanitize=address,undefined for the host code into build/variants/ and runs `pytest -m "not gpu"` against it.  File readers (GGUF,
# ggml, ONNX), the grammar, the tokenizer, the reasoner / decision parser, the fusion and the audio-pipeline state machine all run here.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -I$ROOT/include -I. -Wall -Wno-unused-function -fsanitize=address,undefined -fno-omit-frame-pointer"
make -C "$ROOT/trackiellm_amd/csrc" -j8 OBJDIR="$ROOT/build/obj_asan" OUT="$ROOT/build/variants/lib_asan.so" FLAGS="$FLAGS" HIPCC="/opt/rocm/bin/hipcc -fsanitize=address,undefined" > "$ROOT/build/asan_build.log" 2>&1
cd "$ROOT"
TK_MI355X_LIB="$ROOT/build/variants/lib_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee build/asan_tests.log | tail -5
if grep -q "runtime error\|AddressSanitizer" build/asan_tests.log; then echo "sanitizer reports found: build/asan_tests.log"; exit 1; fi
echo "no sanitizer reports"
