#!/bin/bash
# Round-6 evidence run (on the GPU box, from the repo root), everything from the binary as committed: GPU parity tests, the
# default bench line, rocprofv3 kernel-trace statistics of the same command, PMC passes (one counter group per pass, kernel
# trace only), phase tables. Output: gpurun_out/r06f/ -> copied to profiles/r06_* by tools/r06_collect.py (which also binds
# the PMC traffic summary to the library's SHA-256).
set -u
export TMPDIR=/tmp
out=gpurun_out/r06f
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $out/pytest_gpu.log 2>&1
  tail -4 $out/pytest_gpu.log
fi
if [ "${SKIP_BENCH:-0}" != 1 ]; then
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.log
  tail -1 $out/bench.json | cut -c1-300
fi
B="--no-cpu-baseline --no-shard --no-peaky --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 $B > $out/stats.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 $B --batch 512 > $out/stats512.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_512.csv \;
rm -rf $out/stats.d
timeout 600 bash tools/pmc_run.sh $out fetch_4096 "FETCH_SIZE" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out write_4096 "WRITE_SIZE" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq1_4096 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq2_4096 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" --no-shard --no-peaky --no-extras
CTCDEC_BEAM_KERNEL=wave timeout 300 python bench.py --phases $B --steps 3 > $out/phases4096.json 2> $out/phases4096.log
grep "phase ticks" $out/phases4096.log
CTCDEC_BEAM_KERNEL=wave timeout 300 python bench.py --batch 512 --phases $B --steps 3 > $out/phases512.json 2> $out/phases512.log
grep "phase ticks" $out/phases512.log
ls -la $out
# round 6: the corrected instruction-cost microbenchmark, the exactness check of the short division, when the waves of the launch finish
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip 2>/dev/null && timeout 300 /tmp/valu_rates > $out/micro_valu_rates.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mem_latency tools/micro/mem_latency.hip 2>/dev/null && timeout 120 /tmp/mem_latency > $out/micro_mem_latency.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/np_div_check tools/micro/np_div_check.hip 2>/dev/null && timeout 300 /tmp/np_div_check > $out/np_div_check.txt 2>&1
for p in none dyn weigh; do
  CTCDEC_WAVE_PRIO=$p CTCDEC_WAVE_TIMES=$out/wt_$p.bin timeout 300 python bench.py $B --steps 2 --warmup 1 > /dev/null 2> $out/wt_$p.log
done
python tools/wave_times.py $out/wt_none.bin $out/wt_dyn.bin $out/wt_weigh.bin > $out/wave_times.txt 2>&1
rm -f $out/wt_none.bin* $out/wt_dyn.bin* $out/wt_weigh.bin*
timeout 300 python tools/host_step.py 2>&1 | grep -E "^step|ctcdec host" > $out/host_step.log
tail -3 $out/np_div_check.txt
# the memory paths of the CU (texture addresser / L1 / L2 request counters), one group per pass
timeout 400 bash tools/pmc_run.sh $out ta_4096 "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE" --no-shard --no-peaky --no-extras
timeout 400 bash tools/pmc_run.sh $out tcp_4096 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --no-shard --no-peaky --no-extras
