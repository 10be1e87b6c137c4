#!/bin/bash
set -u
export TMPDIR=/tmp
out=gpurun_out/r06i
mkdir -p $out
timeout 1500 python -m pytest tests/test_probs_golden.py tests/test_golden_full.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1
tail -5 $out/pytest.log
timeout 900 python tools/ab_bench.py --steps 5 "CTCDEC_PRUNE_EXP=pk" "" "CTCDEC_PRUNE_EXP=pk" "" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
timeout 300 python tools/prune_shapes_bench.py > $out/prune_shapes.log 2>&1; tail -20 $out/prune_shapes.log
