"""Diagnostics (GPU box): the time-sliced host ingest with its copy thread against the copies in the after-launch hook, and slice
counts -- 512 utterances of the bench batch as ONE host [B, T, V] array (pageable).   python tools/host_slices_ab.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import bench  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
xs = bench.make_batch(lm, labels, 0, 512, 1000, 6.0, 32)
dec = build_ctcdecoder(labels, lm.path)
dev = torch.from_numpy(xs).cuda()
ref = dec.decode_batch(None, dev, beam_width=100, hotwords=hot)
del dev
for cfg in ({}, {"CTCDEC_HOST_COPY_THREAD": "0"}, {"CTCDEC_HOST_SLICES": "8"}, {"CTCDEC_HOST_SLICES": "16"}, {"CTCDEC_HOST_SLICES": "32"},
            {"CTCDEC_HOST_SLICES": "16", "CTCDEC_HOST_COPY_THREAD": "0"}, {"CTCDEC_HOST_SLICES": "0"}, {}):
    os.environ.update(cfg)
    texts = dec.decode_batch(None, xs, beam_width=100, hotwords=hot)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        texts = dec.decode_batch(None, xs, beam_width=100, hotwords=hot)
        ts.append(1e3 * (time.perf_counter() - t))
    print("HOST %-60s median %.2f ms  min %.2f ms  %.2f M frames/s  texts %s" % (cfg or "default", float(np.median(ts)), min(ts),
                                                                                  512e3 / min(ts) / 1e3, "equal" if texts == ref else "DIFFER"), flush=True)
    for k in cfg:
        os.environ.pop(k)
