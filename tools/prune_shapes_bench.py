"""Diagnostics (GPU box): frame-prune time for label counts around the 64-rows-per-wave kernel's shapes (round 4: any count
up to 2048, aligned or not).  python tools/prune_shapes_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyctcdecode_amd import build_ctcdecoder  # noqa: E402

n, T = 512, 1000
DENSE = os.environ.get("PRUNE_SHAPES_DENSE") == "1"  # flat N(0,1) logits: every label of a small vocabulary survives (configs[1])
LIB = os.environ.get("PRUNE_SHAPES_LIB")             # another build of the library (tools/build_variant.py)
if LIB:
    from pyctcdecode_amd import _binding as B

    B._LIB = B.Library(LIB)
for V in [int(v) for v in sys.argv[1:]] or (1024, 1025, 1027, 1280, 2044, 2047, 2048, 2052, 29, 32):
    dec = build_ctcdecoder([chr(0x4E00 + i) for i in range(V - 1)])
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, T, V), device="cuda", generator=g) * 1.0
    if not DENSE:
        x[:, :, 0] += 6.0  # (a confident blank: short survivor lists, a cheap beam stage)
    for kern in ("fast", "row"):
        if kern == "row":
            os.environ["CTCDEC_PRUNE_KERNEL"] = "row"
        else:
            os.environ.pop("CTCDEC_PRUNE_KERNEL", None)
        bw = 2 if DENSE else 4
        dec.decode_batch(None, x[:64] if DENSE else x, beam_width=bw)
        ts = []
        for _ in range(3):
            dec.decode_batch(None, x[:64] if DENSE else x, beam_width=bw)
            ts.append(dec.last_timing_ms[0] * (n / 64.0 if DENSE else 1.0))
        gb = 4.0 * V * n * T / 1e9
        print("V=%-5d %-4s prune %.3f ms = %.2f TB/s" % (V, kern, min(ts), gb / min(ts)), flush=True)
    os.environ.pop("CTCDEC_PRUNE_KERNEL", None)
    del x
