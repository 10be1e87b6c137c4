#!/bin/bash
# Round-6 diagnostics (GPU box, repo root): what the parts of frame_prune_fast cost -- builds without phase B, without the LDS
# exchange / numpy-order accumulation, with round 5's polynomial in place of numpy's exponential (results are WRONG in these
# builds: timing only).
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06t}
mkdir -p $out
V=pyctcdecode_amd/variants
timeout 1200 python tools/ab_bench.py --steps 6 "" "lib=$V/libctcdec_skipb.so" "lib=$V/libctcdec_noexch.so" "lib=$V/libctcdec_skipb_noexch.so" "lib=$V/libctcdec_pkexp.so" "CTCDEC_PRUNE_EXP=pk" "" 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
