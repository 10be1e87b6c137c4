#!/bin/bash
# Round-3 first device run of the two-waves-per-SIMD wave kernel: parity tests, A/B timings in one process, score gaps
# of the two frame_prune exponentials, phase table. Everything lands under gpurun_out/r03a/.
set -u
export TMPDIR=/tmp
out=gpurun_out/r03a
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -3 $out/pytest_gpu.log
timeout 600 python tools/ab_bench.py --steps 3 "CTCDEC_BEAM_KERNEL=wave" "CTCDEC_PRUNE_EXP=pk" "CTCDEC_BEAM_KERNEL=wave,n=2048" \
  "CTCDEC_BEAM_KERNEL=wave,n=1024" "CTCDEC_BEAM_KERNEL=group,n=1024" "CTCDEC_BEAM_KERNEL=wave,n=512" "CTCDEC_BEAM_KERNEL=group,n=512" \
  "CTCDEC_BEAM_KERNEL=wave,n=256" "CTCDEC_BEAM_KERNEL=group,n=256" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
timeout 300 python tools/golden_full_gap.py > $out/gap_f64exp.log 2>&1; tail -1 $out/gap_f64exp.log
CTCDEC_PRUNE_EXP=pk timeout 300 python tools/golden_full_gap.py > $out/gap_pkexp.log 2>&1; tail -1 $out/gap_pkexp.log
CTCDEC_BEAM_KERNEL=wave timeout 300 python bench.py --batch 512 --phases --no-shard --no-peaky --no-cpu-baseline --steps 3 > $out/phases512.json 2> $out/phases512.log
grep "phase ticks" $out/phases512.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky > $out/stats.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
head -8 $out/kernel_stats_4096.csv
