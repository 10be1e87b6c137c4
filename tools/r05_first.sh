#!/bin/bash
# Round-5 first GPU call (repo root on the GPU box): smoke, the new full-occupancy parity test, the default bench line.
set -u
export TMPDIR=/tmp
out=gpurun_out/r05a
mkdir -p $out
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 1500 python -m pytest tests/test_full_occupancy.py -m gpu -x -q -s --durations=12 > $out/occupancy.log 2>&1; echo "occupancy rc=$?"; tail -25 $out/occupancy.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc=$?"; cat $out/bench.json | head -c 3000
