#!/bin/bash
set -u
mkdir -p gpurun_out/r05m
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/micro/valu_rates.hip 2>/dev/null && timeout 90 /tmp/valu_rates > gpurun_out/r05m/valu_rates.txt 2>&1
cat gpurun_out/r05m/valu_rates.txt
