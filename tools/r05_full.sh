#!/bin/bash
# Round-5 full check (GPU box, repo root): the whole -m gpu suite, then the default bench line
set -u
export TMPDIR=/tmp
out=gpurun_out/r05f${TAG:-}
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench.json"))
keep={k:d[k] for k in ("value","ms_per_step","stages_ms")}
for k in ("shard_512","peaky_posteriors","streaming_cfg5","ragged","fp16_logits","config2","config3","single_utterance","single_real","host_numpy_input_512"):
    v=d.get(k)
    if isinstance(v,dict):
        keep[k]={a:b for a,b in v.items() if a not in ("workload","cpu_baseline","note","sample")}
print(json.dumps(keep,indent=1)[:6000])
print("roofline",d["roofline"]["frac"],d["roofline_frame_prune"]["frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
