"""Diagnostics (GPU box, repo root): one decode() per real-posterior-like 371 x 29 utterance (bench.py's `single_real` leg) --
wall time next to the native call and the stage timings, and a Python profile of the calls.  python tools/single_real_probe.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import synth  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

lm3, xs = bench.inputs_single_real(os.path.join(ROOT, "bench_cache"))
dec = build_ctcdecoder(synth.LIBRI_LABELS, lm3.path)
dev = [torch.from_numpy(x).cuda() for x in xs]
for kernel in (None, "wave", "group"):
    if kernel:
        os.environ["CTCDEC_BEAM_KERNEL"] = kernel
    else:
        os.environ.pop("CTCDEC_BEAM_KERNEL", None)
    dec.decode(dev[0], beam_width=100)
    wall, pr, bm, nat = [], [], [], []
    for rep in range(3):
        for x in dev:
            t0 = time.perf_counter()
            dec.decode(x, beam_width=100)
            wall.append(1e3 * (time.perf_counter() - t0))
            p, b, n = dec.last_timing_ms
            pr.append(p)
            bm.append(b)
            nat.append(n)
    print("SR kernel=%-6s wall %.3f ms  native call %.3f  prune %.3f  beam(+text) %.3f  => python %.3f, native outside the kernels %.3f" % (
        kernel or "auto", np.median(wall), np.median(nat), np.median(pr), np.median(bm), np.median(wall) - np.median(nat),
        np.median(nat) - np.median(pr) - np.median(bm)), flush=True)
os.environ.pop("CTCDEC_BEAM_KERNEL", None)
prof = cProfile.Profile()
prof.enable()
for rep in range(5):
    for x in dev:
        dec.decode(x, beam_width=100)
prof.disable()
pstats.Stats(prof).sort_stats("tottime").print_stats(12)
