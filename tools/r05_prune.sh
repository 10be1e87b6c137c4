#!/bin/bash
# Round-5 prune-kernel check (GPU box, repo root): parity tests that exercise frame_prune_fast, then A/B timing of library variants
set -u
export TMPDIR=/tmp
out=gpurun_out/r05p${TAG:-}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_probs_golden.py tests/test_golden_full.py -m gpu -x -q -k "${KEXPR:-prune or surviv or rows64 or probs or full_size or half or fp16 or bf16 or shapes}" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
timeout 600 python tools/ab_bench.py --steps 5 "$@" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
