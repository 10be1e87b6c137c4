#!/bin/bash
# Round-5 A/B of library variants in one process (GPU box, repo root): tools/build_variant.py builds them, tools/ab_bench.py runs them
set -u
export TMPDIR=/tmp
out=gpurun_out/r05ab${TAG:-}
mkdir -p $out
if [ "${MICRO:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip 2>/dev/null && timeout 120 /tmp/valu_rates > $out/valu_rates.txt 2>&1
  tail -8 $out/valu_rates.txt
fi
timeout 900 python tools/ab_bench.py --steps 5 "$@" > $out/ab.log 2>&1
grep "^AB" $out/ab.log
