#!/bin/bash
# Round-6 experiment (GPU box, repo root): where do the waves of the 4096-utterance launch finish, and does steering the
# issue priority among the waves of a SIMD (CTCDEC_WAVE_PRIO=none|rot<k>|dyn) even them out? Output: gpurun_out/r06b/
set -u
export TMPDIR=/tmp
out=gpurun_out/r06b
mkdir -p $out
B="--no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10"
for p in ${MODES:-none rot5 dyn none dyn rot0}; do
  CTCDEC_WAVE_PRIO=$p timeout 300 python bench.py $B > $out/bench_$p.json 2> $out/bench_$p.log
  echo "prio=$p $(grep 'ms/step' $out/bench_$p.log | tail -1)"
done
for p in ${TIMES:-none dyn}; do
CTCDEC_WAVE_PRIO=$p CTCDEC_WAVE_TIMES=$out/wt_$p.bin timeout 300 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/wt_$p.log
done
python tools/wave_times.py $out/wt_*.bin | tee $out/wave_times.txt
if [ "${SHARD:-1}" = 1 ]; then
for p in none dyn rot5; do
  CTCDEC_WAVE_PRIO=$p timeout 300 python bench.py --batch 1024 --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10 2>&1 >/dev/null | grep 'ms/step' | tail -1 | sed "s/^/batch1024 prio=$p /"
  CTCDEC_WAVE_PRIO=$p timeout 300 python bench.py --batch 2048 --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10 2>&1 >/dev/null | grep 'ms/step' | tail -1 | sed "s/^/batch2048 prio=$p /"
done
fi
