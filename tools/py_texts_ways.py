"""Diagnostics (GPU box): ways of turning a decode_batch result into 4 096 str objects, timed on the real result."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyctcdecode_amd import _binding as B  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
xs = bench.make_batch(lm, labels, 0, 4096, 1000, 6.0, 32)
dec = build_ctcdecoder(labels, lm.path)
dev = torch.from_numpy(xs).cuda()
del xs
params = dec._params(bench.BEAM, -10.0, -5.0, True, 10.0, 1)
params.texts_only = 1
acc = {"texts_of (C loop, PyUnicode_DecodeUTF8 per text)": [], "joined + C split": [], "joined + bytes.decode + str.split": [],
       "joined + bytes.decode only": []}
for it in range(12):
    res = dec._run(dev, params, hot)
    t0 = time.perf_counter()
    a = B.texts_of(dec._lib, res)
    t1 = time.perf_counter()
    blob_p, nbytes, n = C.c_void_p(), C.c_int64(), C.c_int64()
    dec._lib.check(dec._lib.dll.ctcdec_result_texts_joined(res, dec._texts_sep, C.byref(blob_p), C.byref(nbytes), C.byref(n)))
    t2 = time.perf_counter()
    b = B.split_texts(blob_p, int(nbytes.value), int(n.value), dec._texts_sep)
    t3 = time.perf_counter()
    raw = C.string_at(blob_p, int(nbytes.value))
    s = raw.decode("ascii")
    t4 = time.perf_counter()
    c = s.split(dec._texts_sep.decode())
    t5 = time.perf_counter()
    assert a == b == c
    dec._lib.dll.ctcdec_result_free(res)
    if it >= 2:
        acc["texts_of (C loop, PyUnicode_DecodeUTF8 per text)"].append(t1 - t0)
        acc["joined + C split"].append(t3 - t1)
        acc["joined + bytes.decode + str.split"].append((t2 - t1) + (t5 - t3))
        acc["joined + bytes.decode only"].append((t2 - t1) + (t4 - t3))
for k, v in acc.items():
    print("WAYS %-55s median %.3f ms  min %.3f ms" % (k, 1e3 * float(np.median(v)), 1e3 * min(v)), flush=True)
