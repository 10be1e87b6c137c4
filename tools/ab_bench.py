"""Diagnostics (GPU box): time decode_batch of the bench workload under several switch settings inside ONE process
(same resident logits, same decoder), so that variants are compared on the same box and the same data.
  python tools/ab_bench.py [--batch 4096] [--steps 3] CONFIG [CONFIG ...]
CONFIG = comma-separated KEY=VALUE environment switches and/or n=<utterances>, bw=<beam width>, lib=<path of another build of
the library: tools/build_variant.py>, e.g.
  "CTCDEC_BEAM_KERNEL=wave,n=4096"  "CTCDEC_BEAM_KERNEL=group,n=512"  "CTCDEC_PRUNE_EXP=pk"
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--kind", default="words", choices=["words", "peaky"])
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
    xs = bench.make_batch(lm, labels, 0, args.batch, args.frames, 6.0, min(64, os.cpu_count() or 1), args.kind)
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    from pyctcdecode_amd import _binding as B

    decs = {"": build_ctcdecoder(labels, lm.path)}
    cur_lib = ""
    default_lib = B.get_library()
    dev = torch.from_numpy(xs).cuda()
    torch.cuda.synchronize()
    del xs
    ref_texts = {}
    for cfg in args.configs:
        env, n, bw, lib = {}, args.batch, bench.BEAM, ""
        for kv in cfg.split(","):
            if not kv:
                continue
            k, v = kv.split("=", 1)
            if k == "n":
                n = int(v)
            elif k == "bw":
                bw = int(v)
            elif k == "lib":
                lib = v
            else:
                env[k] = v
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        if lib != cur_lib:  # another build of the library: its own decoder (the previous one is freed first: workspaces are big)
            decs.clear()
            B._LIB = B.Library(lib) if lib else default_lib
            decs[lib] = build_ctcdecoder(labels, lm.path)
            cur_lib = lib
        dec = decs[cur_lib]
        try:
            batch = dev[:n]
            texts = dec.decode_batch(None, batch, beam_width=bw, hotwords=hot)  # warm-up
            wall, pr, bm = [], [], []
            for _ in range(args.steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                texts = dec.decode_batch(None, batch, beam_width=bw, hotwords=hot)
                wall.append(1000 * (time.perf_counter() - t0))
                pr.append(dec.last_timing_ms[0])
                bm.append(dec.last_timing_ms[1])
            same = None
            if (n, bw) in ref_texts:
                same = texts == ref_texts[(n, bw)]
            else:
                ref_texts[(n, bw)] = texts
            print("AB %-60s n=%-5d wall %.2f ms (min %.2f)  prune %.2f  beam %.2f  kernel %s  %.1f Mframes/s%s" % (
                cfg, n, float(np.median(wall)), min(wall), float(np.median(pr)), float(np.median(bm)),
                bench.KERNEL_NAMES.get(dec.last_beam_kernel, "?"), n * args.frames / min(wall) / 1e3,
                "" if same is None else ("  texts==first" if same else "  TEXTS DIFFER")), flush=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


if __name__ == "__main__":
    main()
