"""Diagnostics (GPU box): where the host side of one decode_batch step goes -- native call vs. packing the texts vs.
Python objects.  python tools/host_tail.py [batch]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pyctcdecode_amd import _binding as B  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
xs = bench.make_batch(lm, labels, 0, n, 1000, 6.0, 32)
dec = build_ctcdecoder(labels, lm.path)
dev = torch.from_numpy(xs).cuda()
os.environ["CTCDEC_HOST_TIMING"] = "1"
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    params = dec._params(100, -10.0, -5.0, True, 10.0, 1)
    res = dec._run(dev, params, hot)
    t1 = time.perf_counter()
    pk = B.Packed()
    dec._lib.check(dec._lib.dll.ctcdec_result_pack(res, C.byref(pk)))
    t2 = time.perf_counter()
    nb = int(pk.n_beams)
    text_off = np.ctypeslib.as_array(pk.text_off, shape=(nb + 1,))
    blob = C.string_at(pk.text_blob, int(text_off[nb]))
    off = text_off.tolist()
    text = blob.decode("ascii")
    texts = [text[off[k]:off[k + 1]] for k in range(nb)]
    t3 = time.perf_counter()
    dec._lib.dll.ctcdec_result_free(res)
    t4 = time.perf_counter()
    print("batch %d: run %.2f ms (native %.2f: prune %.2f beam %.2f), pack %.2f ms, python strings %.2f ms, free %.2f ms" % (
        n, 1e3 * (t1 - t0), dec.last_timing_ms[2], dec.last_timing_ms[0], dec.last_timing_ms[1], 1e3 * (t2 - t1), 1e3 * (t3 - t2),
        1e3 * (t4 - t3)), flush=True)
