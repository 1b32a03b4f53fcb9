#!/bin/bash
# builds tools/lds_trace/libddp_emu_trace.so: the emulator with every memory access hooked (see hooks.cpp)
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd); D=$ROOT/tools/lds_trace
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O1 -g -std=c++17 -fPIC -ffp-contract=off -w -fno-vectorize -fno-slp-vectorize -fno-unroll-loops -DDDP_EMU_TRACE \
  -fsanitize=kernel-address -mllvm -asan-instrumentation-with-call-threshold=0 -mllvm -asan-stack=0 -mllvm -asan-globals=0 -mllvm -asan-opt=0 \
  -c $ROOT/tests/emu/emu.cpp -o $D/emu_trace.o
$CXX -O2 -g -std=c++17 -fPIC -c $D/hooks.cpp -o $D/hooks.o
$CXX -shared -o $D/libddp_emu_trace.so $D/emu_trace.o $D/hooks.o -ldl
echo $D/libddp_emu_trace.so
