#!/bin/bash
# Round-5 working measurement (GPU box, repo root): headline step + instruction counters of the same binary. Output: gpurun_out/r05q/
set -u
export TMPDIR=/tmp
out=gpurun_out/r05q${TAG:-}
mkdir -p $out
if [ "${MICRO:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip 2>/dev/null && /tmp/valu_rates > $out/valu_rates.txt 2>&1
  cat $out/valu_rates.txt
fi
timeout 600 python bench.py --batch ${BATCH:-4096} --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 10 > $out/bench.json 2> $out/bench.log
grep "ms/step" $out/bench.log | tail -1
if [ "${SKIP_PMC:-0}" != 1 ]; then
timeout 600 bash tools/pmc_run.sh $out sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky --no-extras
python - <<PY
import json
try:
    d=json.load(open("$out/sq1.json"))
    d=d.get("kernels",d)
    for k,v in d.items():
        if k.startswith("beam_wave") or k.startswith("frame_prune_fast"):
            n = 4096*1000 if k.startswith("beam_wave") else 4096*1000
            print(k,{a:(round(b/n,1) if isinstance(b,float) else b) for a,b in v.items()})
except Exception as e:
    print("sq1 missing",e)
PY
fi
if [ "${PHASES:-0}" = 1 ]; then
CTCDEC_BEAM_KERNEL=wave timeout 300 python bench.py --batch ${BATCH:-4096} --phases --no-shard --no-peaky --no-cpu-baseline --no-extras --steps 3 > $out/phases.json 2> $out/phases.log
grep "phase ticks" $out/phases.log
fi
