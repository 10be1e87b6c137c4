"""Diagnostics (GPU box): largest |score - reference score| of the HIP build over the full-size golden cases, per input dtype
(tests/golden/cases_full.json.gz; the tests bound it at 1e-4).  python tools/golden_full_gap.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyctcdecode_amd import build_ctcdecoder  # noqa: E402
from tests import test_golden_full as G  # noqa: E402

cache = os.path.join(ROOT, "bench_cache")
assets = G.bench.build_assets(cache, 20000, 60000)
gap = {}
decs = {}
for case in G.CASES:
    labels, arpa, x, kw = G._input(case, assets)
    key = (id(labels), arpa)
    if key not in decs:
        decs[key] = build_ctcdecoder(labels, arpa)
    out = decs[key].decode_beams(torch.from_numpy(x).cuda(), **kw)
    g = 0.0
    for o, e in zip(out, case["expected"]):
        if o.text == e["text"]:
            g = max(g, abs(o.lm_score - e["lm"]), abs(o.logit_score - e["logit"]))
    gap[case["dtype"]] = max(gap.get(case["dtype"], 0.0), g)
    print("%-22s %-8s max gap %.3e" % (case["name"], case["dtype"], g), flush=True)
print("max gap per input dtype:", {k: "%.2e" % v for k, v in gap.items()})
