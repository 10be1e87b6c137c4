#!/bin/bash
# Round-3 third device run: parity tests incl. device-resident streams, the full default bench line (all secondary lines).
set -u
export TMPDIR=/tmp
out=gpurun_out/r03c
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -5 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.log
grep "bench " $out/bench.log | tail -25
tail -1 $out/bench.json | cut -c1-600
