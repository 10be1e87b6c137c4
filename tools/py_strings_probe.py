"""Diagnostics (GPU box): how long the Python side of decode_batch takes per step over 30 steps (4 096 str objects of ~1 KB from the
library's text blocks), under the allocator settings given in the environment.   python tools/py_strings_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
xs = bench.make_batch(lm, labels, 0, 4096, 1000, 6.0, 32)
dec = build_ctcdecoder(labels, lm.path)
dev = torch.from_numpy(xs).cuda()
del xs
torch.cuda.synchronize()
out, wall = [], []
texts = None
for it in range(32):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    texts = dec.decode_batch(None, dev, beam_width=bench.BEAM, hotwords=hot)
    t1 = time.perf_counter()
    wall.append(1e3 * (t1 - t0))
    out.append(1e3 * (t1 - t0) - dec.last_timing_ms[2])
o, w = np.array(out[2:]), np.array(wall[2:])
print("PY %-70s wall median %.2f mean %.2f max %.2f ms; outside the native call: median %.3f mean %.3f max %.3f ms" % (
    " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith(("MALLOC_", "CTCDEC_TEXT"))) or "defaults",
    np.median(w), w.mean(), w.max(), np.median(o), o.mean(), o.max()), flush=True)
