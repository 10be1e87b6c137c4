#!/bin/bash
# Round-6 (GPU box, repo root): after the same-address atomics left utt_sniff / utt_weigh -- kernel statistics, the host side of a
# step (tools/host_tail.py), 512 / 1024 / 2048-utterance launches under both priority modes.
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06m}
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 600 python tools/ab_bench.py --steps 8 "" CTCDEC_WAVE_PRIO=dyn "" "n=2048" "n=2048,CTCDEC_WAVE_PRIO=dyn" "n=1024" "n=1024,CTCDEC_WAVE_PRIO=dyn" "n=1025" "n=1025,CTCDEC_WAVE_PRIO=weigh" 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
timeout 300 python tools/host_tail.py 4096 2>&1 | grep -E "^batch|ctcdec host" | tee $out/host_tail.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky --no-extras > $out/stats.log 2>&1
find $out/stats.d -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
head -12 $out/kernel_stats_4096.csv
grep "ms/step" $out/stats.log | tail -1
