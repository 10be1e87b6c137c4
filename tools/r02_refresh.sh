#!/bin/bash
# Lighter re-run of tools/r02_profile.sh after a kernel change: GPU tests, the bench line, kernel-trace stats and the
# 4096-utterance PMC passes (the 512-utterance PMC files of the full run stay as they are).
set -u
export TMPDIR=/tmp
out=gpurun_out/r02
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -3 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.log
tail -1 $out/bench.json | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky > $out/stats.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky --batch 512 > $out/stats512.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_512.csv \;
rm -rf $out/stats.d
timeout 600 bash tools/pmc_run.sh $out fetch_4096 "FETCH_SIZE" --no-shard --no-peaky --batch 4096
timeout 600 bash tools/pmc_run.sh $out write_4096 "WRITE_SIZE" --no-shard --no-peaky --batch 4096
timeout 600 bash tools/pmc_run.sh $out sq1_4096 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky
timeout 600 bash tools/pmc_run.sh $out sq2_4096 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" --no-shard --no-peaky
ls $out | wc -l
