"""PCIe-inclusive rate of the headline workload: the same batch handed over as HOST numpy matrices
(staged H2D inside the call) instead of a device tensor. Never the reported bench value."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401  (one HIP runtime per process)

import bench  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

lm, labels, hot = bench.build_assets(os.path.join(sys.path[0], "bench_cache"), 20000, 60000)
xs = bench.make_batch(lm, labels, 0, 512, 1000)
dec = build_ctcdecoder(labels, lm.path)
for mode in ("pageable list of [T,V]", "one [B,T,V] array"):
    arg = xs if mode.startswith("pageable") else np.stack(xs)
    dec.decode_batch(None, arg, beam_width=100, hotwords=hot)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        dec.decode_batch(None, arg, beam_width=100, hotwords=hot)
        ts.append(time.perf_counter() - t)
    dt = min(ts)
    print("host input (%s): %.1f ms/step, %.2f M frames/s, kernels %.2f+%.2f ms" % (
        mode, dt * 1e3, 512e3 / dt / 1e6, dec.last_timing_ms[0], dec.last_timing_ms[1]))
