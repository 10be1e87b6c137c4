"""Diagnostics (GPU box, from the repo root): one T=1000 utterance of the headline workload through decode() and decode_beams()
(with and without history pruning; C and Python result building) -- wall time next to the stage timings."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
import torch
from pyctcdecode_amd import build_ctcdecoder
lm, labels, hot = bench.build_assets("bench_cache", 20000, 60000)
xs = torch.from_numpy(bench.make_batch(lm, labels, 0, 4, 1000, 6.0, 4, "words")).cuda()
dec = build_ctcdecoder(labels, lm.path)
def t(fn, n=12):
    fn()
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts)), r
for name, fn in (("decode", lambda: dec.decode(xs[0], hotwords=hot)),
                 ("decode_beams ph=False", lambda: dec.decode_beams(xs[0], hotwords=hot)),
                 ("decode_beams ph=True", lambda: dec.decode_beams(xs[0], hotwords=hot, prune_history=True))):
    ms, r = t(fn)
    print("SP %-24s %.2f ms  timing(prune, beam, native) %s  n=%s" % (name, ms, tuple(round(v, 2) for v in dec.last_timing_ms), len(r) if isinstance(r, list) else 1), flush=True)
os.environ["CTCDEC_PY_UNPACK"] = "1"
ms, r = t(lambda: dec.decode_beams(xs[0], hotwords=hot))
print("SP decode_beams python unpack %.2f ms" % ms)
