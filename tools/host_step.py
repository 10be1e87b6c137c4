"""Diagnostics (GPU box): the host side of the bench step -- decode_batch on the resident headline batch, wall time against the
native call and its kernels (CTCDEC_HOST_TIMING breaks the native call down).   python tools/host_step.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
xs = bench.make_batch(lm, labels, 0, n, 1000, 6.0, 32)
dec = build_ctcdecoder(labels, lm.path)
dev = torch.from_numpy(xs).cuda()
torch.cuda.synchronize()
os.environ["CTCDEC_HOST_TIMING"] = "1"
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    texts = dec.decode_batch(None, dev, beam_width=bench.BEAM, hotwords=hot)
    t1 = time.perf_counter()
    ms = dec.last_timing_ms
    print("step %d: wall %.3f ms, native %.3f ms, kernels %.3f + %.3f = %.3f ms; python outside the native call %.3f ms" % (
        it, 1e3 * (t1 - t0), ms[2], ms[0], ms[1], ms[0] + ms[1], 1e3 * (t1 - t0) - ms[2]), flush=True)

# the Python side of the call, piece by piece (the same calls decode_batch makes)
import ctypes as C  # noqa: E402

from pyctcdecode_amd import _binding as B  # noqa: E402
from pyctcdecode_amd.decoder import _Batch, _c_arrays  # noqa: E402

for it in range(3):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    params = dec._params(bench.BEAM, -10.0, -5.0, True, 10.0, 1)
    params.texts_only = 1
    dec._set_hotwords(hot)
    t.append(time.perf_counter())
    batch = _Batch(dev, len(dec._idx2vocab))
    ptrs, frames = _c_arrays(batch)
    t.append(time.perf_counter())
    res = C.c_void_p()
    dec._lib.check(dec._lib.dll.ctcdec_decode_batch(dec._handle, ptrs, frames, len(batch.ptrs), batch.dtype, int(batch.is_device),
                                                    C.byref(params), None, C.byref(res)))
    t.append(time.perf_counter())
    texts = B.texts_of(dec._lib, res)
    t.append(time.perf_counter())
    dec._lib.dll.ctcdec_result_free(res)
    t.append(time.perf_counter())
    d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
    print("python pieces %d: params + hot words %.3f ms, batch object + pointer arrays %.3f, native call (ctypes) %.3f, %d str objects %.3f, "
          "result freed %.3f" % (it, d[0], d[1], d[2], len(texts), d[3], d[4]), flush=True)
