#!/bin/bash
# Round-6 (GPU box, repo root): the placement test, when the waves of the default launch end, the weights of the last ones.
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06q}
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 900 python -m pytest tests/test_full_occupancy.py -m gpu -q -x -k placement 2>&1 | tail -3 | tee $out/pytest_placement.log
B="--no-shard --no-peaky --no-cpu-baseline --no-extras"
for p in weigh dyn; do
  CTCDEC_WAVE_PRIO=$p CTCDEC_WAVE_TIMES=$out/wt_$p.bin timeout 300 python bench.py $B --steps 2 --warmup 1 > /dev/null 2> $out/wt_$p.log
done
python tools/wave_times.py $out/wt_weigh.bin $out/wt_dyn.bin | tee $out/wave_times.txt
