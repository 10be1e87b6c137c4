#!/bin/bash
# Round-4 working measurement (GPU box, repo root): phase ticks of the wave kernel at full occupancy and the SQ counters of
# the same binary. Output: gpurun_out/r04q/
set -u
export TMPDIR=/tmp
out=gpurun_out/r04q
mkdir -p $out
CTCDEC_BEAM_KERNEL=wave timeout 300 python bench.py --batch ${BATCH:-4096} --phases --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 3 > $out/phases.json 2> $out/phases.log
grep "phase ticks" $out/phases.log
grep "ms/step" $out/phases.log | tail -1
if [ "${SKIP_PMC:-0}" != 1 ]; then
timeout 600 bash tools/pmc_run.sh $out sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq2 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" --no-shard --no-peaky --no-extras
python - <<'PY'
import json
for t in ("sq1","sq2"):
    try:
        d=json.load(open("gpurun_out/r04q/%s.json"%t))
    except Exception as e:
        print(t,"missing",e); continue
    for k,v in d.items():
        if k.startswith("beam_wave"):
            print(t,k,{a:(round(b/4096/1000,1) if isinstance(b,float) else b) for a,b in v.items()})
PY
fi
