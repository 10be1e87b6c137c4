#!/bin/bash
# usage: tools/pmc_run.sh <out_dir> <tag> "<COUNTER ...>" [bench args...]   (run on the GPU box, from the repo root)
# One rocprofv3 --pmc pass of bench.py (kernel trace only, as gpurun requires), summarised into <out_dir>/<tag>.json
set -u
out=$1; tag=$2; ctrs=$3; shift 3
export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/$tag.d" -o "$tag" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/$tag.log" 2>&1
python tools/pmc_summary.py "$out/$tag.d" "$out/$tag.json" > /dev/null
rm -rf "$out/$tag.d"
