#!/bin/bash
# Round-6 experiment (GPU box, repo root): issue priority by PREDICTED work (CTCDEC_WAVE_PRIO=weigh: utt_weigh / utt_place,
# backend_hip.hip) against priority by frames left (dyn): step time A/B in one process, wave end times, and how well the
# weight predicts a wave's natural lifetime. Output: gpurun_out/r06j/
set -u
export TMPDIR=/tmp
out=gpurun_out/r06j
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 900 python tools/ab_bench.py --steps 8 CTCDEC_WAVE_PRIO=dyn CTCDEC_WAVE_PRIO=weigh CTCDEC_WAVE_PRIO=weigh,CTCDEC_NO_PLACE=1 \
  CTCDEC_WAVE_PRIO=dyn CTCDEC_WAVE_PRIO=weigh CTCDEC_WAVE_PRIO=none 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
for p in none dyn weigh; do
  CTCDEC_WAVE_PRIO=$p CTCDEC_WAVE_TIMES=$out/wt_$p.bin timeout 300 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/wt_$p.log
done
CTCDEC_WAVE_PRIO=weigh CTCDEC_NO_PLACE=1 CTCDEC_WAVE_TIMES=$out/wt_weigh_noplace.bin timeout 300 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2> $out/wt_weigh_noplace.log
python tools/wave_times.py $out/wt_*.bin | tee $out/wave_times.txt
