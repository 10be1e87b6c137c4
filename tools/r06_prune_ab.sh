#!/bin/bash
# Round-6 (GPU box, repo root): where the exact float32 prune stage spends its time -- diagnostic variants (wrong sums on purpose)
set -u
export TMPDIR=/tmp
out=gpurun_out/r06h
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/np_div_check tools/micro/np_div_check.hip 2>/dev/null && timeout 300 /tmp/np_div_check | tee $out/np_div_check.txt
timeout 900 python tools/ab_bench.py --steps 5 "CTCDEC_PRUNE_EXP=pk" "" "lib=pyctcdecode_amd/variants/libctcdec_npdiag_pkexp.so" "lib=pyctcdecode_amd/variants/libctcdec_npdiag_noexch.so" "lib=pyctcdecode_amd/variants/libctcdec_npdiag_both.so" "" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
timeout 600 bash tools/pmc_run.sh $out sq_np "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out sq_np2 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" --no-shard --no-peaky --no-extras
for f in $out/sq_np.json $out/sq_np2.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); d=d.get("kernels",d)
for k,v in d.items():
    if k.startswith("frame_prune_fast"):
        print(k[:44],{a:(round(b/4096e3,2) if isinstance(b,(int,float)) else b) for a,b in v.items()})
PY
done
