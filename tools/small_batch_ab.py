"""Diagnostics (GPU box): the two beam kernels on the small-batch BASELINE configs (config2: 256 x T=1000, V=29, no LM, D_flat;
config3: 512 x T=1000, V=32, 4-gram) and on 256 / 512 utterances of the headline workload."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch

    cache = os.path.join(ROOT, "bench_cache")
    for kernel in ("group", "wave"):
        os.environ["CTCDEC_BEAM_KERNEL"] = kernel
        for name, fn in (("config2", lambda: bench.extra_config2(torch, 3)), ("config3", lambda: bench.extra_config3(torch, cache, 3))):
            r = fn()
            print("SB %-8s %-6s %.2f ms/step  beam %.2f ms  (%s)" % (name, kernel, r["ms_per_step"], r["stages_ms"]["beam"], r["kernel"]), flush=True)


if __name__ == "__main__":
    main()
