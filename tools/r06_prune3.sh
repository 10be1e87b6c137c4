#!/bin/bash
# Round-6 (GPU box, repo root): frame_prune_fast with 48 / 32 rows per wave and four waves per SIMD (LDS 9 / 6 KB, 128 registers)
# against the shipped 64 rows / three waves; one process, same resident batch (tools/ab_bench.py lib=...).
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06o}
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so pyctcdecode_amd/variants/*.so > $out/library.sha256
V=pyctcdecode_amd/variants
timeout 1200 python tools/ab_bench.py --steps 8 "" "lib=$V/libctcdec_r48w4.so" "lib=$V/libctcdec_r64w4.so" "lib=$V/libctcdec_r32w4.so" "lib=$V/libctcdec_r64w3il1.so" "" "lib=$V/libctcdec_r48w4.so" 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
