#!/bin/bash
# Runs the emulated-kernel tests (the product's HIP kernel sources compiled for the CPU wave
# emulator) under UBSan and ASan.  GPU sanitizers are not available on the pool, so this is where
# out-of-bounds accesses and undefined behaviour in the kernels get caught: every global / LDS access
# of a kernel is a real host access here.  usage: tools/sanitize_emu.sh [ubsan|asan]
set -e
cd "$(dirname "$0")/../tests/hipemu"
mode=${1:-ubsan}
mkdir -p _san
CXXF="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -march=x86-64-v3 -Wno-unknown-pragmas -I."
if [ "$mode" = asan ]; then
  g++ $CXXF -fsanitize=address -shared -o _san/libpicsong_emu_asan.so emu_driver.cpp emu_runtime.cpp
  RT=$(gcc -print-file-name=libasan.so); SO=$PWD/_san/libpicsong_emu_asan.so
  # lanes run on malloc'ed coroutine stacks: no stack-use-after-return tracking, no leak check of python
  export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1
else
  g++ $CXXF -fsanitize=undefined -fno-sanitize-recover=undefined -shared -o _san/libpicsong_emu_ubsan.so emu_driver.cpp emu_runtime.cpp
  RT=$(gcc -print-file-name=libubsan.so); SO=$PWD/_san/libpicsong_emu_ubsan.so
  export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
fi
cd ../..
LD_PRELOAD=$RT PICSONG_EMU_SO=$SO python -m pytest tests/test_kernels_emulated.py -x -q
