#!/bin/bash
# Round-6 experiment (GPU box, repo root): where do the waves of the 4096-utterance launch finish, and does rotating the
# issue priority among the waves of a SIMD (CTCDEC_WAVE_PRIO=<shift>) even them out? Output: gpurun_out/r06b/
set -u
export TMPDIR=/tmp
out=gpurun_out/r06b
mkdir -p $out
B="--no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10"
for p in none 5 2 0 8 none 5; do
  if [ $p = none ]; then unset CTCDEC_WAVE_PRIO; else export CTCDEC_WAVE_PRIO=$p; fi
  timeout 300 python bench.py $B > $out/bench_$p.json 2> $out/bench_$p.log
  echo "prio=$p $(grep 'ms/step' $out/bench_$p.log | tail -1)"
done
unset CTCDEC_WAVE_PRIO
CTCDEC_WAVE_TIMES=$out/wt_none.bin timeout 300 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/wt_none.log
CTCDEC_WAVE_PRIO=5 CTCDEC_WAVE_TIMES=$out/wt_5.bin timeout 300 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/wt_5.log
CTCDEC_WAVE_PRIO=0 CTCDEC_WAVE_TIMES=$out/wt_0.bin timeout 300 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/wt_0.log
python tools/wave_times.py $out/wt_none.bin $out/wt_5.bin $out/wt_0.bin | tee $out/wave_times.txt
