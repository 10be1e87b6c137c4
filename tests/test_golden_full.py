"""Parity at FULL SIZE against the unmodified reference: tests/golden/cases_full.json.gz holds what /root/reference
returned (oracle/make_golden_full.py) for utterances of the bench workload itself -- T=1000, V=1024, 20k-word 4-gram
LM, hot words -- fed as float64 and as float32, and for one BASELINE config-2 utterance: EVERY returned beam (up to 100
per case) with its word frames, and every one is compared. Inputs are regenerated from their seeds.
tests/golden/cases_peaky.json.gz (oracle/make_golden_peaky.py) adds real-posterior-like utterances of the same
vocabulary / LM -- the inputs on which the kernels consume runs of single-label frames in place.

All bounds are ABSOLUTE (scores are around -2000 here):
* float64 inputs: order / frames exact, |score - reference| <= 1e-9 (measured 4.6e-13 on the device), near-tie window 1e-9.
* float32 inputs: the reference runs _log_softmax in the INPUT dtype (decoder.py:180-197). Round 6: so does the product --
  numpy's float32 exp / log and its summation order restated (csrc/np_f32.h, np_sum.h): the same bound as float64, 1e-9
  (measured: 0 on the simulator). Under CTCDEC_PRUNE_EXP=pk / f64 (the round-5 / round-2 routines, fp64 from an exact
  upcast) the scores differ from the reference's by the reference's own fp32 rounding (1.8e-5 over T=1000): bound 1e-4,
  order exact wherever the reference's scores are further apart than 4e-5.
"""
import gzip
import json
import os

import numpy as np
import pytest

import bench
import synth
from tests.golden_util import GOLD, check_beams
from tests.sim_util import sim_library  # noqa: F401

with gzip.open(os.path.join(GOLD, "cases_full.json.gz"), "rt", encoding="utf-8") as f:
    FULL = json.load(f)
with gzip.open(os.path.join(GOLD, "cases_peaky.json.gz"), "rt", encoding="utf-8") as f:  # real-posterior-like inputs
    PEAKY = json.load(f)["cases"]
CASES = FULL["cases"] + PEAKY
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def assets():
    cache = os.path.join(ROOT, "bench_cache") if os.access(ROOT, os.W_OK) else "/tmp/ctc_bench"
    return bench.build_assets(cache, 20000, 60000)


def _input(case, assets):
    lm, labels, hot = assets
    if case["kind"] == "config2":
        return synth.LIBRI_LABELS, None, synth.d_flat(2, case["utt"], case["frames"], 29).astype(case["dtype"]), dict(case["decode"])
    if case["kind"] == "peaky":
        x = synth.d_peaky(bench.CONFIG_ID + 1, case["utt"], bench.T, labels, True, lm.words, lm.sentences, len(labels),
                          boost=case["boost"], unsure=case["unsure"])
    else:
        x = synth.d_words(bench.CONFIG_ID, case["utt"], bench.T, labels, True, lm.words, lm.sentences, len(labels), boost=6.0)
    kw = dict(case["decode"])
    kw["hotwords"] = hot if kw["hotwords"] == "bench" else None
    return labels, lm.path, x.astype(case["dtype"]), kw


def _f32_exact():
    return os.environ.get("CTCDEC_PRUNE_EXP", "np")[0] == "n"


def _check(case, got, tol):
    """got: [(text, frames, logit, lm)] of ALL returned beams."""
    assert len(got) == case["n_beams"], "%s: %d beams, the reference returned %d" % (case["name"], len(got), case["n_beams"])
    exp = case["expected"]
    assert len(exp) == case["n_beams"]  # (round 4: every beam is committed)
    tie = 1e-9 if (case["dtype"] == "float64" or _f32_exact()) else 4e-5
    check_beams(got, exp, tol=tol, what=case["name"], tie_tol=tie)  # texts, word frames, both scores, order -- all beams


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("bench_u0_float64", "bench_u1_float32", "bench_u3_float32",
                                                                     "peaky_u1_float32", "peaky_u2_float64")],
                         ids=lambda c: c["name"])
def test_oracle_equals_the_reference_at_full_size(case, assets):
    """The oracle keeps the input dtype like the reference: float32 cases must agree to 1e-9 as well."""
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd.alphabet import Alphabet

    labels, arpa, x, kw = _input(case, assets)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe, arpa, None)
    with np.errstate(all="ignore"):
        out = orc.decode_beams(x, **kw)
    assert len(out) == case["n_beams"]
    check_beams([(o[0], o[2], o[3], o[4]) for o in out], case["expected"], tol=1e-9, what=case["name"])


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("bench_u0_float64", "bench_u1_float32") or c["kind"] == "peaky"],
                         ids=lambda c: c["name"])
def test_sim_equals_the_reference_at_full_size(case, assets, sim_library, both_beam_kernels):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    labels, arpa, x, kw = _input(case, assets)
    dec = build_ctcdecoder(labels, arpa)
    out = dec.decode_beams(x, **kw)
    _check(case, [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out],
           1e-9 if (case["dtype"] == "float64" or _f32_exact()) else 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_hip_equals_the_reference_at_full_size(case, assets, both_beam_kernels):
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    labels, arpa, x, kw = _input(case, assets)
    dec = build_ctcdecoder(labels, arpa)
    xt = torch.from_numpy(x).cuda()  # the device tensor in its own dtype: fp32 logits are read in place
    out = dec.decode_beams(xt, **kw)
    got = [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out]
    _check(case, got, 1e-9 if (case["dtype"] == "float64" or _f32_exact()) else 1e-4)
    gap = max(abs(g[3] - e["lm"]) for g, e in zip(got, case["expected"]) if g[0] == e["text"])
    print("%s [%s]: max |lm_score - reference| = %.3g" % (case["name"], both_beam_kernels, gap))
