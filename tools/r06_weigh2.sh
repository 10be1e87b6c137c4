#!/bin/bash
# Round-6 experiment, second run (GPU box, repo root): the placement kernels after utt_place was spread over the device.
set -u
export TMPDIR=/tmp
out=gpurun_out/r06k
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 900 python tools/ab_bench.py --steps 8 CTCDEC_WAVE_PRIO=dyn CTCDEC_WAVE_PRIO=weigh CTCDEC_WAVE_PRIO=weigh,CTCDEC_NO_PLACE=1 \
  CTCDEC_WAVE_PRIO=dyn CTCDEC_WAVE_PRIO=weigh 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky --no-extras > $out/stats.log 2>&1
find $out/stats.d -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
head -12 $out/kernel_stats_4096.csv
