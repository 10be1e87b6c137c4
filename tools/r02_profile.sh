#!/bin/bash
# Round-2 evidence run (on the GPU box, from the repo root): GPU parity tests, the default bench line, a
# rocprofv3 kernel-trace of the same command and the PMC passes (one counter group per pass, kernel trace only).
# Everything lands under gpurun_out/r02/; the summaries are copied into profiles/ by hand afterwards.
set -u
export TMPDIR=/tmp
out=gpurun_out/r02
mkdir -p $out
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
  tail -3 $out/pytest_gpu.log
fi
timeout 600 python bench.py > $out/bench.json 2> $out/bench.log
tail -1 $out/bench.json | cut -c1-400
# 1-GPU rehearsal of the N>1 path (process group, device binding, the one all_gather per step)
CTC_BENCH_FORCE_DIST=1 timeout 300 python bench.py --batch 1024 --steps 3 > $out/bench_dist1.json 2> $out/bench_dist1.log
tail -1 $out/bench_dist1.json | cut -c1-200
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky > $out/stats.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky --batch 512 > $out/stats512.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_512.csv \;
rm -rf $out/stats.d
for b in 4096 512; do
  timeout 600 bash tools/pmc_run.sh $out fetch_$b "FETCH_SIZE" --no-shard --no-peaky --batch $b
  timeout 600 bash tools/pmc_run.sh $out write_$b "WRITE_SIZE" --no-shard --no-peaky --batch $b
done
timeout 600 bash tools/pmc_run.sh $out sq1_4096 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky
timeout 600 bash tools/pmc_run.sh $out sq2_4096 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" --no-shard --no-peaky
ls -la $out
