#!/bin/bash
# Round-5 counter passes on the memory paths of the beam kernel (GPU box, repo root; one group per rocprofv3 pass, kernel trace only)
set -u
export TMPDIR=/tmp
out=gpurun_out/r05pmc
mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA|TCP|TCC|SQ|SQC|TD)_[A-Za-z0-9_]+" | sort -u > $out/counters_avail.txt
wc -l $out/counters_avail.txt
B="--no-shard --no-peaky --no-extras"
timeout 400 bash tools/pmc_run.sh $out sqw "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" $B
timeout 400 bash tools/pmc_run.sh $out sqi "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" $B
timeout 400 bash tools/pmc_run.sh $out ta "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE" $B
timeout 400 bash tools/pmc_run.sh $out tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $B
python - <<PY
import json
for t in ("sqw","sqi","ta","tcp"):
    try:
        d=json.load(open("$out/%s.json"%t))
    except Exception as e:
        print(t,"missing",e); continue
    for k,v in d.items():
        if k.startswith("beam_wave") or k.startswith("frame_prune_fast"):
            print(t,k,{a:(round(b/4096/1000,2) if isinstance(b,float) else b) for a,b in v.items() if a not in ("grid","wg","lds","scratch","sgpr","vgpr","dispatches")})
PY
tail -3 $out/ta.log; tail -3 $out/tcp.log
