"""Diagnostics (GPU box): the workgroup kernel with 256 and with 512 threads per utterance (CTCDEC_GROUP_THREADS=512, taken
when every CU holds at most one utterance) on BASELINE config 2 (256 x T=1000, V=29, no LM, D_flat: ~2 500 candidates per
frame) and on 256 utterances of the headline workload (a few dozen candidates per frame)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import synth  # noqa: E402


def main():
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    os.environ["CTCDEC_BEAM_KERNEL"] = "group"
    dec2 = build_ctcdecoder(synth.LIBRI_LABELS)
    x2 = torch.from_numpy(np.stack([synth.d_flat(2, u, 1000, 29) for u in range(256)])).cuda()
    lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
    xh = torch.from_numpy(bench.make_batch(lm, labels, 0, 256, 1000, 6.0, 32, "words")).cuda()
    dech = build_ctcdecoder(labels, lm.path)
    ref = {}
    for threads in ("256", "512", "256", "512"):
        os.environ["CTCDEC_GROUP_THREADS"] = threads
        for name, dec, x, kw in (("config2", dec2, x2, {}), ("headline256", dech, xh, {"hotwords": hot})):
            texts, dt, prune, beam = bench._time_decode_batch(torch, dec, x, 3, beam_width=100, **kw)
            same = ref.setdefault(name, texts) == texts
            print("G5 %-12s threads %s: %.2f ms/step, beam %.2f ms, texts equal %s" % (name, threads, 1e3 * dt, beam, same), flush=True)


if __name__ == "__main__":
    main()
