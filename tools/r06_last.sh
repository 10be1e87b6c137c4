#!/bin/bash
# Round-6 closing run (GPU box, repo root) on the FINAL tree, after tools/r06_final.sh's counter passes were collected and
# committed (bench.py binds roofline.traffic to them): the whole GPU suite and the default bench line with every leg.
set -u
export TMPDIR=/tmp
out=gpurun_out/r06f
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library_last.sha256
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $out/pytest_gpu.log 2>&1
tail -6 $out/pytest_gpu.log
timeout 1200 python bench.py > $out/bench.json 2> $out/bench.log
tail -1 $out/bench.json | cut -c1-400
FUZZ_BACKEND=hip timeout 900 python tools/fuzz_sim_vs_oracle.py 250 606 > $out/fuzz_hip.log 2>&1; tail -3 $out/fuzz_hip.log
