#!/bin/bash
# Round-6 (GPU box, repo root): time-sliced host ingest -- parity tests, then the bench line with its host-input leg
set -u
export TMPDIR=/tmp
out=gpurun_out/r06g
mkdir -p $out
timeout 1500 python -m pytest tests/test_host_slices.py tests/test_kenlm_binary.py tests/test_np_f32.py -m gpu -q -x > $out/pytest_host.log 2>&1
tail -6 $out/pytest_host.log
CTCDEC_SLICE_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-peaky --steps 10 > $out/bench.json 2> $out/bench.log
grep -E "host_numpy|time-sliced|ms/step|float32_polynomial|shard" $out/bench.log | tail -12
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06g/bench.json").read().strip().splitlines()[-1])
for k in ("host_numpy_input_512","float32_polynomial_mode","shard_512"):
    print(k, json.dumps(d.get(k))[:600])
PY
CTCDEC_HOST_SLICES=0 timeout 600 python bench.py --no-cpu-baseline --no-peaky --no-extras --steps 5 2>&1 >/dev/null | grep -E "host_numpy|ms/step" | tail -3
