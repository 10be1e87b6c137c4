"""ORACLE PINNING of the input sniff (this container only): outputs of the UNMODIFIED reference (/root/reference through
oracle/refshim/) on float32 / float16 PROBABILITY matrices.

The reference reads its input as probabilities when  math.isclose(logits.sum(axis=1).mean(), 1)  (decoder.py:760) -- on
the INPUT dtype. A float32 softmax output has row sums like 0.99999994 / 1 / 1.0000001; whether their float32 mean is
EXACTLY 1 (the neighbours of 1 are 6e-8 and 1.2e-7 away, the tolerance is 1e-9) decides between  log(clip(p))  and
log_softmax(p)  -- two very different decodes. This script searches seeds for both outcomes and stores, per case, the
input matrix itself (numpy's float32 exp is not bit-identical across CPUs, so the inputs are NOT regenerated) and what
the reference returned.

    python oracle/make_golden_probs.py       (seconds)

-> tests/golden/cases_probs.json + tests/golden/inputs_probs.npz
"""
import json
import math
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference", ROOT]
os.chdir("/tmp")

import logging  # noqa: E402
import warnings  # noqa: E402

import numpy as np  # noqa: E402

logging.disable(logging.CRITICAL)
warnings.simplefilter("ignore")

from pyctcdecode import build_ctcdecoder  # noqa: E402  (the reference)

import synth  # noqa: E402
from tests.golden_util import TOY_ARPA  # noqa: E402

TOY_LABELS = [" ", "b", "g", "n", "s", "u", "y", ""]


def softmax_in(dtype, rng, T, V, scale, via16):
    """A float32 softmax (how a model hands probabilities over); via16: one that went through float16 on the way (row sums
    off by ~1e-4: NOT probabilities to the reference's test although they are to the eye)."""
    x = (rng.standard_normal((T, V)) * scale).astype(np.float32)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    p = e / e.sum(axis=1, keepdims=True)
    if via16:
        p = p.astype(np.float16)
    return p.astype(dtype)


def find(dtype, want_prob, T, V, scale, start, via16):
    """First seed >= start whose matrix the reference's test classifies as wanted."""
    for seed in range(start, start + 20000):
        x = softmax_in(dtype, np.random.default_rng(seed), T, V, scale, via16)
        if math.isclose(x.sum(axis=1).mean(), 1) == want_prob:
            return seed, x
    raise RuntimeError("no seed found")


def beams_json(beams):
    return [{"text": b.text, "frames": [[w, int(f[0]), int(f[1])] for w, f in b.text_frames],
             "logit": float(b.logit_score), "lm": float(b.lm_score)} for b in beams]


def main():
    plan = [  # (name, labels, arpa, dtype, is_prob wanted, T, scale, decode kwargs)
        ("libri_f32_prob_a", "libri", None, "float32", True, 60, 3.0, {}),
        ("libri_f32_prob_b", "libri", None, "float32", True, 97, 2.0, {"prune_history": True}),
        ("libri_f32_notprob_tiny", "libri", None, "float32", False, 2, 3.0, {}),          # two rows: 0.99999994 is not 1
        ("libri_f32_notprob_via16", "libri", None, "float32", False, 97, 2.0, {"prune_history": True}),
        ("libri_f32_prob_via16", "libri", None, "float32", True, 7, 2.0, {}),             # ... and sometimes it is
        ("toy_f32_prob_lm", "toy", "toy", "float32", True, 40, 2.0, {"beam_width": 16}),
        ("toy_f32_notprob_lm", "toy", "toy", "float32", False, 3, 2.0, {"beam_width": 16}),
        ("libri_f16_prob", "libri", None, "float16", True, 50, 3.0, {}),
        ("libri_f16_notprob_tiny", "libri", None, "float16", False, 2, 3.0, {}),
    ]
    cases, arrays = [], {}
    start = 0
    for name, lab, arpa, dtype, want, T, scale, kw in plan:
        labels = synth.LIBRI_LABELS if lab == "libri" else TOY_LABELS
        ref = build_ctcdecoder(list(labels), TOY_ARPA if arpa == "toy" else None)
        V = len(labels) + (0 if "" in labels else 1)
        seed, x = find(np.dtype(dtype).type, want, T, V, scale, start, "via16" in name)
        start = seed + 1
        with np.errstate(all="ignore"):
            beams = ref.decode_beams(x, **kw)
        rs = x.sum(axis=1)
        cases.append({"name": name, "labels": lab, "lm": arpa, "dtype": dtype, "is_prob": want, "seed": seed, "decode": kw,
                      "mean_row_sum": repr(float(rs.mean())), "expected": beams_json(beams)})
        arrays[name] = x
        ref.cleanup()
        print(name, "seed", seed, "mean row sum", repr(float(rs.mean())), len(beams), "beams", repr(beams[0].text[:40]), file=sys.stderr)
    with open(os.path.join(ROOT, "tests", "golden", "cases_probs.json"), "w") as f:
        json.dump({"cases": cases}, f, ensure_ascii=False)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "inputs_probs.npz"), **arrays)
    print("wrote tests/golden/cases_probs.json, inputs_probs.npz", file=sys.stderr)


if __name__ == "__main__":
    main()
