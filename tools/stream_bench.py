"""Diagnostics (GPU box): BASELINE configs[4] -- 64 streams x 50-frame chunks, beam 200 -- chunk by chunk, with the host-side
timing of the native call (CTCDEC_HOST_TIMING=1).  python tools/stream_bench.py [streams] [chunk] [beam]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    beam = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
    xs = bench.make_batch(lm, labels, 0, n, 1000, 6.0, min(32, os.cpu_count() or 1), "words")
    import torch

    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.language_model import HotwordScorer

    dec = build_ctcdecoder(labels, lm.path)
    dev = torch.from_numpy(xs).cuda()
    n_chunks = 1000 // chunk
    chunks = [dev[:, k * chunk:(k + 1) * chunk].contiguous() for k in range(n_chunks)]
    scorer = HotwordScorer.build_scorer(hot, weight=10.0)
    torch.cuda.synchronize()
    for rep in range(3):
        if rep == 2:
            os.environ["CTCDEC_HOST_TIMING"] = "1"
        states = [dec.get_starting_state() for _ in range(n)]
        c1s, c2s = [s[1] for s in states], [s[2] for s in states]
        beams = [s[0] for s in states]
        ts = []
        for k in range(n_chunks - 1):
            t0 = time.perf_counter()
            beams = dec.partial_decode_beams_batch(chunks[k], c1s, c2s, beams, [k * chunk] * n, beam_width=beam, hotword_scorer=scorer,
                                                   prune_history=True)
            ts.append(1e3 * (time.perf_counter() - t0))
        print("rep %d: per-chunk ms: median %.3f min %.3f max %.3f; first five %s" % (
            rep, float(np.median(ts)), min(ts), max(ts), ["%.2f" % t for t in ts[:5]]), flush=True)
    os.environ.pop("CTCDEC_HOST_TIMING", None)


if __name__ == "__main__":
    main()
