#!/bin/bash
# Round-5 closing run (GPU box, repo root): the whole GPU suite and the default bench line on the final tree (final binary +
# the last Python-side changes); profiles/r05_pmc_hbm_traffic.json is bound to this binary, so the line carries roofline.traffic.
set -u
export TMPDIR=/tmp
out=gpurun_out/r05f
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc=$?"
tail -1 $out/bench.json | cut -c1-200
