"""Diagnostics (GPU box): BASELINE configs[2] (V=32 character labels, 4-gram, beam 100, 512 x T=1000) under both beam kernels and
both workgroup sizes, with the phase ticks of each.  python tools/config3_ab.py [utterances]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (puts tests/ on the path)
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402
from tools.small_batch_ab import G_NAMES, W_NAMES  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    lm_u, labels, xs_np = bench.inputs_config3(os.path.join(ROOT, "bench_cache"))
    dec = build_ctcdecoder(labels, lm_u.path, alpha=0.5, beta=1.0)
    xs = torch.from_numpy(xs_np[:n]).cuda()
    ref = None
    for env in ({"CTCDEC_BEAM_KERNEL": "group"}, {"CTCDEC_BEAM_KERNEL": "group", "CTCDEC_GROUP_THREADS": "512"},
                {"CTCDEC_BEAM_KERNEL": "wave"}):
        for k in ("CTCDEC_BEAM_KERNEL", "CTCDEC_GROUP_THREADS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        t = dec.decode_batch(None, xs, beam_width=100)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t = dec.decode_batch(None, xs, beam_width=100)
            ts.append(1e3 * (time.perf_counter() - t0))
        same = "" if ref is None else (" same" if t == ref else " DIFFER")
        ref = ref or t
        print("CFG3 n=%d %s wall %.2f ms beam %.2f kernel %s%s" % (n, env, min(ts), dec.last_timing_ms[1],
              bench.KERNEL_NAMES.get(dec.last_beam_kernel), same), flush=True)
        dll = dec._lib.dll
        dll.ctcdec_profile_phases(dec._handle, 1, None, 0)
        dec.decode_batch(None, xs, beam_width=100)
        ticks = (C.c_uint64 * 24)()
        dll.ctcdec_profile_phases(dec._handle, 0, ticks, 24)
        names = W_NAMES if dec.last_beam_kernel == 1 else G_NAMES
        tot = float(sum(ticks)) or 1.0
        print("CFG3   phases: " + ", ".join("%s %.0f us (%.0f%%)" % (nm, tk / 100.0, 100.0 * tk / tot)
                                            for nm, tk in zip(names, ticks) if tk), flush=True)
