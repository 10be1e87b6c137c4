#!/bin/bash
# Round-6 (GPU box, repo root): what of utt_place / the weights carries the gain -- the weights' gain in the priority rule, the
# snake, the order itself. One process, same resident batch.
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06p}
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 1200 python tools/ab_bench.py --steps 8 "" CTCDEC_WEIGH_GAIN=0 CTCDEC_WEIGH_GAIN=3 CTCDEC_WEIGH_GAIN=6 CTCDEC_PLACE_SNAKE=0 CTCDEC_PLACE_SNAKE=2 "CTCDEC_WEIGH_GAIN=3,CTCDEC_PLACE_SNAKE=0" "" CTCDEC_WEIGH_GAIN=3 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
