#!/bin/bash
# Round-6 (GPU box, repo root): the whole GPU suite on the current tree, the headline step, the host-copy probe
set -u
export TMPDIR=/tmp
out=gpurun_out/r06e
mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $out/pytest_gpu.log 2>&1
tail -12 $out/pytest_gpu.log
B="--no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10"
for m in np pk np; do
  if [ $m = np ]; then unset CTCDEC_PRUNE_EXP; else export CTCDEC_PRUNE_EXP=$m; fi
  timeout 300 python bench.py $B > $out/bench_$m.json 2> $out/bench_$m.log
  echo "exp=$m $(grep 'ms/step' $out/bench_$m.log | tail -1)"
done
unset CTCDEC_PRUNE_EXP
timeout 300 python tools/h2d_2d_probe.py 2>&1 | tee $out/h2d_probe.txt
