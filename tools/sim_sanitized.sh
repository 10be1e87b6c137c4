#!/bin/bash
# The CPU simulator of the beam kernels (tests/sim) built with AddressSanitizer + UBSan, and the stress suites run on it:
# out-of-bounds LDS / arena accesses that a GPU would swallow silently abort here. From the repo root:
#   bash tools/sim_sanitized.sh [pytest args, default: the simulator suites]
set -eu
out=tests/_build/libctcdec_sim_asan.so
mkdir -p tests/_build
g++ -std=c++17 -O1 -g -fPIC -shared -pthread -DCTC_SIM -DCTC_TEXT_WIN=48 -DCTC_TEXT_LIST=16 -fsanitize=address,undefined \
    -fno-omit-frame-pointer -Wno-unused-function -o $out pyctcdecode_amd/csrc/api.cpp pyctcdecode_amd/csrc/host_tables.cpp pyctcdecode_amd/csrc/kenlm_binary.cpp \
    tests/sim/backend_sim.cpp
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export CTCDEC_SIM_LIB=$PWD/$out FUZZ_LIB=$PWD/$out
if [ $# -gt 0 ]; then
  python -m pytest "$@"
else
  python -m pytest tests/test_sim_vs_oracle.py tests/test_sim_golden.py tests/test_sim_streaming.py tests/test_golden_full.py \
      tests/test_multi_lm.py -q -m "not gpu" -x
  python tools/fuzz_sim_vs_oracle.py 200 9402
fi
