#!/bin/bash
# Round-5 last GPU minutes: on the box, is the shipped library still "built" by build()'s content stamp, and does the default
# workload's bench line bind the committed PMC summary (roofline.traffic filled, traffic_source naming the binding)?
set -u
out=gpurun_out/r05g
mkdir -p $out
timeout 60 python -c "
from pyctcdecode_amd import build as b
import hashlib
print('needs_build', b.needs_build(), 'source_tag', b.source_tag()[:16], 'library', hashlib.sha256(open(b.OUT,'rb').read()).hexdigest()[:16])
" > $out/bind_check.txt 2>&1
cat $out/bind_check.txt
timeout 110 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard --no-peaky --no-extras > $out/bench_short.json 2> $out/bench_short.log; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05g/bench_short.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"))
except Exception as e:
    print("no line:", e)
PY
