"""The input sniff in the INPUT dtype (decoder.py:759-765): float32 / float16 probability matrices.

tests/golden/cases_probs.json + inputs_probs.npz hold what the unmodified reference returned (oracle/make_golden_probs.py)
for softmax outputs whose float32 (float16) mean row sum is exactly 1 -- read as probabilities, log(clip(p)) -- and for
ones where it is not -- read as logits, log_softmax(p) --, including the cases a float64 evaluation of the test gets wrong
(`*_via16`: float16-rounded probabilities whose float32 mean is 1 although their exact mean is 1 - 1.6e-5, and tiny
utterances whose rows sum to 0.99999994).

* float32: the reference keeps float32 all the way in the probability branch (numpy's weak Python scalars: np.log, the
  clip and every score stay float32), the device computes in float64 from the same float32 values: scores within 1e-4
  ABSOLUTE (measured <= 1.4e-5), order exact outside runs of reference scores closer than 4e-5.
* float16: the reference's own arithmetic is float16 there (scores above 32 are multiples of 1/32 .. 1/8, log(0) = -inf
  because the clip bound 1e-15 is 0 in float16) -- not something to reproduce. What is pinned is the classification: the
  product must equal the oracle run on the exact float64 upcast with the sniff taken on the float16 matrix.
"""
import json
import os

import numpy as np
import pytest

import synth
from tests.golden_util import GOLD, TOY_ARPA, check_beams
from tests.sim_util import sim_library  # noqa: F401

with open(os.path.join(GOLD, "cases_probs.json")) as f:
    CASES = json.load(f)["cases"]
INPUTS = np.load(os.path.join(GOLD, "inputs_probs.npz"))
TOY_LABELS = [" ", "b", "g", "n", "s", "u", "y", ""]


def _setup(case):
    labels = synth.LIBRI_LABELS if case["labels"] == "libri" else TOY_LABELS
    return list(labels), (TOY_ARPA if case["lm"] else None), INPUTS[case["name"]]


def _oracle(labels, arpa):
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd.alphabet import Alphabet

    alpha = Alphabet.build_alphabet(labels)
    return build_oracle(alpha.labels, alpha.is_bpe, arpa, None)


def _check_product(case, decode):
    labels, arpa, x = _setup(case)
    got = [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in decode(labels, arpa, x, case["decode"])]
    if case["dtype"] == "float32" and not case["is_prob"] and os.environ.get("CTCDEC_PRUNE_EXP", "np")[0] == "n":
        # read as logits: the log-softmax in the reference's own float32 arithmetic (np_f32.h), fp64 from there on like the reference
        check_beams(got, case["expected"], tol=1e-9, what=case["name"])
    elif case["dtype"] == "float32":
        # read as probabilities the reference keeps float32 through every SCORE (numpy's weak Python scalars): the float32 bound
        check_beams(got, case["expected"], tol=1e-4, tie_tol=4e-5, what=case["name"])
    else:
        with np.errstate(all="ignore"):
            exp = _oracle(labels, arpa).decode_beams(x.astype(np.float64), sniff_on=x, **case["decode"])
        check_beams(got, [{"text": e[0], "frames": [[w, int(a), int(b)] for w, (a, b) in e[2]], "logit": e[3], "lm": e[4]}
                          for e in exp], what=case["name"])


@pytest.mark.parametrize("case", [c for c in CASES if c["dtype"] == "float32"], ids=lambda c: c["name"])
def test_oracle_takes_the_reference_branch(case):
    """The oracle restates decoder.py:759-765 on the input dtype: same branch as the reference, scores within the float32
    noise of the reference's own probability branch."""
    labels, arpa, x = _setup(case)
    with np.errstate(all="ignore"):
        out = _oracle(labels, arpa).decode_beams(x, **case["decode"])
    check_beams([(o[0], o[2], o[3], o[4]) for o in out], case["expected"], tol=1e-4, tie_tol=4e-5, what=case["name"])


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_sim_reads_probabilities_like_the_reference(case, sim_library, both_beam_kernels):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    _check_product(case, lambda labels, arpa, x, kw: build_ctcdecoder(labels, arpa).decode_beams(x, **kw))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_hip_reads_probabilities_like_the_reference(case, both_beam_kernels):
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    # the device tensor in its own dtype (read in place), and the same matrix as a host array
    _check_product(case, lambda labels, arpa, x, kw: build_ctcdecoder(labels, arpa).decode_beams(torch.from_numpy(x).cuda(), **kw))
    _check_product(case, lambda labels, arpa, x, kw: build_ctcdecoder(labels, arpa).decode_beams(x, **kw))
