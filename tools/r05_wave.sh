#!/bin/bash
# Round-5 wave-kernel check (GPU box, repo root): parity tests that run the wave kernel, then A/B timing of library variants
set -u
export TMPDIR=/tmp
out=gpurun_out/r05w${TAG:-}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_full.py tests/test_resident_streams.py tests/test_full_occupancy.py -m gpu -x -q -k "${KEXPR:-wave or occupancy}" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 600 python tools/ab_bench.py --steps 5 "$@" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
