#!/bin/bash
# Round-6 (GPU box, repo root): GPU suite + bench line after the placement kernels and the stage-major exponentials.
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06l}
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 600 python tools/ab_bench.py --steps 8 "" CTCDEC_WAVE_PRIO=dyn "" 2>&1 | grep -E "^AB|Error|error" | tee $out/ab.log
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1800 python -m pytest tests -m gpu -q -x --durations=5 > $out/pytest_gpu.log 2>&1
  tail -4 $out/pytest_gpu.log
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard --no-peaky --no-extras > $out/stats.log 2>&1
find $out/stats.d -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
head -8 $out/kernel_stats_4096.csv
grep "ms/step" $out/stats.log | tail -1
