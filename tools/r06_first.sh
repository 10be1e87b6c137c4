#!/bin/bash
# Round-6 first measurement (GPU box, repo root): the corrected instruction-cost microbenchmark, the memory-path latencies, the
# headline step of the tree as round 5 left it, and the per-class instruction counters of the same binary. Output: gpurun_out/r06a/
set -u
export TMPDIR=/tmp
out=gpurun_out/r06a
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip 2>/dev/null && timeout 300 /tmp/valu_rates > $out/valu_rates.txt 2>&1
echo "valu_rates rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/mem_latency tools/micro/mem_latency.hip 2>/dev/null && timeout 120 /tmp/mem_latency > $out/mem_latency.txt 2>&1
cat $out/valu_rates.txt | cut -c1-250
cat $out/mem_latency.txt
timeout 600 python bench.py --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10 > $out/bench.json 2> $out/bench.log
grep "ms/step" $out/bench.log | tail -1
timeout 600 bash tools/pmc_run.sh $out sq_insts "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq_valu_kinds "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq_valu_kinds2 "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq_busy "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq_thread "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT" --no-shard --no-peaky --no-extras
ls $out
for f in $out/sq_*.json; do echo == $f; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); d=d.get("kernels",d)
for k,v in d.items():
    if k.startswith("beam_wave") or k.startswith("frame_prune_fast"):
        print(k[:40],{a:(round(b/4096e3,2) if isinstance(b,(int,float)) else b) for a,b in v.items()})
PY
done
