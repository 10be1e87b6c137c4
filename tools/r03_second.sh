#!/bin/bash
# Round-3 second device run: parity tests (both kernels x both frame_prune exponentials), A/B timings, SQ counters of the
# new wave kernel, phase table. Output under gpurun_out/r03b/.
set -u
export TMPDIR=/tmp
out=gpurun_out/r03b
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -4 $out/pytest_gpu.log
timeout 600 python tools/ab_bench.py --steps 3 "CTCDEC_BEAM_KERNEL=wave" "CTCDEC_PRUNE_EXP=f64" "CTCDEC_BEAM_KERNEL=wave,n=2048" \
  "CTCDEC_BEAM_KERNEL=wave,n=1024" "CTCDEC_BEAM_KERNEL=wave,n=512" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
CTCDEC_BEAM_KERNEL=wave timeout 300 python bench.py --batch 512 --phases --no-shard --no-peaky --no-cpu-baseline --steps 3 > $out/phases512.json 2> $out/phases512.log
grep "phase ticks" $out/phases512.log
timeout 600 bash tools/pmc_run.sh $out sq1_4096 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky
timeout 600 bash tools/pmc_run.sh $out sq2_4096 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" --no-shard --no-peaky
cat $out/sq1_4096.json $out/sq2_4096.json | head -80
