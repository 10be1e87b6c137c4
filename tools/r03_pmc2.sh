#!/bin/bash
# instruction-cache and stall counters of the wave kernel at two occupancies (512 utterances = 2 per CU, 2048 = 8 per CU)
set -u
export TMPDIR=/tmp
out=gpurun_out/r03d
mkdir -p $out
for b in 512 2048; do
  CTCDEC_BEAM_KERNEL=wave timeout 600 bash tools/pmc_run.sh $out ic_$b "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" --no-shard --no-peaky --no-extras --batch $b
  CTCDEC_BEAM_KERNEL=wave timeout 600 bash tools/pmc_run.sh $out dc_$b "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" --no-shard --no-peaky --no-extras --batch $b
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03d/*.json')):
    d=json.load(open(f))
    for k,v in d.items():
        if 'beam_wave' in k: print(f.split('/')[-1], {a:b for a,b in v.items() if a.startswith('SQ')})
PY
CTCDEC_HOST_TIMING=1 timeout 300 python tools/ab_bench.py --steps 2 "CTCDEC_BEAM_KERNEL=wave" 2>&1 | grep "ctcdec host\|^AB" | tail -4
