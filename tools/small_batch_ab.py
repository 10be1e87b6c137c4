"""Diagnostics (GPU box): wave kernel against workgroup kernel for BASELINE configs[1] (29 labels, ~2 500 candidates a frame)
at several batch sizes, and the phase ticks of each.  python tools/small_batch_ab.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (puts tests/ on the path: synth)
import synth  # noqa: E402
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

G_NAMES = ["load", "modes", "completions(bar)", "keys", "merge", "score(bar)", "clear", "sort.rank", "rebuild.tail",
           "rest", "finalise", "comp.src", "comp.probe", "comp.store", "score.fold", "score.probe", "score.push",
           "sort.zero", "sort.compact", "rebuild.hist", "rebuild.dup", "rebuild.build", "comp.syncmem", "pool.prune"]
W_NAMES = ["modes", "completions", "keys", "match", "fold", "score", "rank", "build.write", "finalise", "compact",
           "push", "tok.load", "build.gather", "big.match+fold", "tables", "label.runs", "st.comp", "st.push", "st.build"] + ["-"] * 5

if __name__ == "__main__":
    dec = build_ctcdecoder(synth.LIBRI_LABELS)
    xs = torch.from_numpy(bench.inputs_config2()).cuda()
    sizes = [int(v) for v in sys.argv[1:]] or [256, 64]
    for n in sizes:
        ref = None
        for kern in ("group", "wave"):
            os.environ["CTCDEC_BEAM_KERNEL"] = kern
            t = dec.decode_batch(None, xs[:n], beam_width=100)
            ts = []
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                t = dec.decode_batch(None, xs[:n], beam_width=100)
                ts.append(1e3 * (time.perf_counter() - t0))
            same = "" if ref is None else (" same" if t == ref else " DIFFER")
            ref = ref or t
            print("CFG2 n=%-4d %-5s wall %.2f ms beam %.2f kernel %s%s" % (n, kern, min(ts), dec.last_timing_ms[1],
                  bench.KERNEL_NAMES.get(dec.last_beam_kernel), same), flush=True)
            dll = dec._lib.dll
            dll.ctcdec_profile_phases(dec._handle, 1, None, 0)
            dec.decode_batch(None, xs[:n], beam_width=100)
            ticks = (C.c_uint64 * 24)()
            dll.ctcdec_profile_phases(dec._handle, 0, ticks, 24)
            names = W_NAMES if dec.last_beam_kernel == 1 else G_NAMES
            tot = float(sum(ticks)) or 1.0
            print("CFG2   phases: " + ", ".join("%s %.0f us (%.0f%%)" % (nm, tk / 100.0, 100.0 * tk / tot)
                                                for nm, tk in zip(names, ticks) if tk), flush=True)
