"""Diagnostics: decode one golden case (or its first T frames) with both beam kernels of the loaded backend and
print where their outputs first differ.  python tools/diff_kernels.py <case name> [sim]"""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
logging.disable(logging.WARNING)
from pyctcdecode_amd import _binding as B  # noqa: E402

if len(sys.argv) > 2 and sys.argv[2] == "sim":
    from tests.sim.build_sim import build

    B._LIB = B.Library(build())
from pyctcdecode_amd import build_ctcdecoder  # noqa: E402
from tests.golden_util import lm_path, load_cases  # noqa: E402

CASES, INPUTS = load_cases()
case = [c for c in CASES if c["name"] == sys.argv[1]][0]
x = INPUTS[case["input"]]


def run(kernel, T):
    os.environ["CTCDEC_BEAM_KERNEL"] = kernel
    dec = build_ctcdecoder(case["labels"], lm_path(case["lm"]), case["unigrams"], **case["build"])
    out = dec.decode_beams(x[:T], **case["decode"])
    return [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out]


for T in range(1, len(x) + 1):
    a, b = run("wave", T), run("group", T)
    if [(p[0], p[1]) for p in a] != [(q[0], q[1]) for q in b]:
        print("first difference at T =", T, ": wave", len(a), "beams, group", len(b))
        for k, (p, q) in enumerate(zip(a, b)):
            if p[:2] != q[:2]:
                print(k, "wave ", p[0][-30:], p[1][-2:], p[2], p[3])
                print(k, "group", q[0][-30:], q[1][-2:], q[2], q[3])
                break
        for k in range(min(8, len(a), len(b))):
            print("   ", k, repr(a[k][0][-20:]), round(a[k][3], 6), "|", repr(b[k][0][-20:]), round(b[k][3], 6))
        break
else:
    print("kernels agree on all", len(x), "prefixes")
