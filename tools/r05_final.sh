#!/bin/bash
# Round-5 last pass on the FINAL binary (GPU box, repo root): the tests around the last host-side change, the default bench line,
# rocprofv3 kernel statistics and the HBM-traffic counter passes that bench.py binds to the library's hash. The SQ / memory-path
# counters and phase tables of tools/r05_profile.sh were taken one host-side rule earlier (same device kernels) and are kept.
set -u
export TMPDIR=/tmp
out=gpurun_out/r05f
mkdir -p $out
sha256sum pyctcdecode_amd/libctcdec.so > $out/library.sha256
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden_full.py -m gpu -x -q -k "small_batches or ragged or edge or bpe1025 or full_size" > $out/pytest_final_subset.log 2>&1; echo "subset rc=$?"; tail -3 $out/pytest_final_subset.log
B="--no-cpu-baseline --no-shard --no-peaky --no-extras"
timeout 600 bash tools/pmc_run.sh $out fetch_4096 "FETCH_SIZE" --no-shard --no-peaky --no-extras
timeout 600 bash tools/pmc_run.sh $out write_4096 "WRITE_SIZE" --no-shard --no-peaky --no-extras
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 $B > $out/stats.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_4096.csv \;
rm -rf $out/stats.d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats.d -o stats -- python bench.py --steps 5 --warmup 2 $B --batch 512 > $out/stats512.log 2>&1
find $out/stats.d -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats_512.csv \;
rm -rf $out/stats.d
timeout 900 python bench.py > $out/bench.json 2> $out/bench.log
tail -1 $out/bench.json | cut -c1-200
