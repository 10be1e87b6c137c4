#!/bin/bash
# Round-6 check (GPU box, repo root): float32 rows in the reference's own float32 arithmetic -- parity suites + what the prune stage costs
set -u
export TMPDIR=/tmp
out=gpurun_out/r06c
mkdir -p $out
timeout 1500 python -m pytest tests/test_probs_golden.py tests/test_golden_full.py tests/test_gpu_parity.py tests/test_np_f32.py -m gpu -q -x --durations=5 > $out/pytest_f32.log 2>&1
tail -15 $out/pytest_f32.log
B="--no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10"
for m in np pk np pk; do
  if [ $m = np ]; then unset CTCDEC_PRUNE_EXP; else export CTCDEC_PRUNE_EXP=$m; fi
  timeout 300 python bench.py $B > $out/bench_$m.json 2> $out/bench_$m.log
  echo "exp=$m $(grep 'ms/step' $out/bench_$m.log | tail -1)"
done
