"""Known answers for the n-gram arithmetic ABOVE order 2, derived by hand from the ARPA definition -- not from any
implementation in this tree. (The reference pins kenlm's behaviour only on its toy 2-gram; the headline
workload is a 4-gram.) Reference call sites: language_model.py:308-360 (LanguageModel.score / get_start_state over
kenlm.Model.BaseScore).

The model, tests/golden/ngram_known.arpa (order 4; every number is a multiple of 2^-5, so fp32 sums are exact):

    1-grams  log10 p   back-off        2-grams            3-grams               4-grams
    <unk>    -1.0       0              <s> a  -0.5   -0.375   <s> a b  -0.3125 -0.0625   <s> a b c  -0.15625
    <s>      -2.0      -0.5            a b    -0.625 -0.25    a b c    -0.4375 -0.1875   a b c a    -0.09375
    </s>     -1.5       0              b c    -0.875 -0.5     b c a    -0.1875  0
    a        -1.25     -0.25           c a    -1.125  0
    b        -1.75     -0.75           b </s> -0.25   0
    c        -2.25     -0.125

ARPA semantics (Katz back-off): log P(w | h) = p(h w) if the n-gram "h w" is listed, else bo(h) + log P(w | h minus its
oldest word), with bo(h) = 0 when "h" is not listed or lists no back-off.

Derivations (h written oldest word first):
  Q1  P(a | <s>)        = p(<s> a)                                               = -0.5          full match, order 2
  Q2  P(b | <s> a)      = p(<s> a b)                                             = -0.3125       full match, order 3
  Q3  P(c | <s> a b)    = p(<s> a b c)                                           = -0.15625      full match, order 4
  Q4  P(a | a b c)      = p(a b c a)                                             = -0.09375      full match, the 4-gram window slid
  Q5  P(c | b c a)      = bo(b c a) + bo(c a) + bo(a) + p(c) = 0 + 0 - 0.25 - 2.25 = -2.5          three back-off steps
  Q6  P(a | <s> a b)    = bo(<s> a b) + bo(a b) + bo(b) + p(a)
                        = -0.0625 - 0.25 - 0.75 - 1.25                           = -2.3125       three steps, all weights non-zero
  Q7  P(</s> | <s> a b) = bo(<s> a b) + bo(a b) + p(b </s>) = -0.0625 - 0.25 - 0.25 = -0.5625      two steps down to a bigram
  Q8  P(zzz | <s> a b)  = bo(<s> a b) + bo(a b) + bo(b) + p(<unk>)
                        = -0.0625 - 0.25 - 0.75 - 1.0                            = -2.0625       OOV in the middle of a context
  Q9  P(a | ... zzz)    = bo(<unk>) + p(a) = 0 - 1.25                            = -1.25         the context after an OOV word is <unk> alone
  Q10 P(a | empty)      = p(a)                                                   = -1.25         NullContext: no <s>
  Q11 no <unk> line (ngram_known_nounk.arpa): kenlm gives the missing <unk> log10 p = -100:
      P(zzz | empty) = -100, P(zzz | a) = bo(a) - 100 = -100.25

State after a query (the convention of oracle/arpa_lm.py and the device tables; kenlm itself may hold a shorter, equivalent
state where no longer n-gram can extend): the words of the longest matched n-gram, newest first, at most order - 1 = 3 of
them, each context length with its listed back-off.
"""
import math
import os

import numpy as np
import pytest

from tests.golden_util import GOLD
from tests.sim_util import sim_library  # noqa: F401

ARPA = os.path.join(GOLD, "ngram_known.arpa")
ARPA_NOUNK = os.path.join(GOLD, "ngram_known_nounk.arpa")

# the tables live in tests/golden/known_answers.json ("ngram_order4"): (name, start state, context words oldest first, word,
# expected log10 p, expected out-state (words newest first, back-offs) or null)
import json  # noqa: E402

with open(os.path.join(GOLD, "known_answers.json")) as _f:
    _KA = json.load(_f)["ngram_order4"]
CHAIN = [tuple(c) for c in _KA["chain"]]
CHAIN_NOUNK = [tuple(c) for c in _KA["chain_without_unk_line"]]


def _run(model, new_state, chain, word_of):
    for name, start, ctx, word, want, want_state in chain:
        st = new_state()
        (model.BeginSentenceWrite if start == "bos" else model.NullContextWrite)(st)
        for w in ctx:
            nxt = new_state()
            model.BaseScore(st, w, nxt)
            st = nxt
        out = new_state()
        got = model.BaseScore(st, word, out)
        assert got == want, (name, got, want)  # exact: every term is a multiple of 2^-5
        if want_state is not None:
            words, backoffs = want_state
            n = len(out.words)
            assert [word_of(i) for i in out.words[:n]] == words, (name, [word_of(i) for i in out.words[:n]], words)
            assert [float(b) for b in out.backoff[:n]] == backoffs, (name, list(out.backoff), backoffs)


def test_oracle_arpa_model_against_hand_derived_answers():
    from oracle.arpa_lm import ArpaModel, ArpaState

    m = ArpaModel(ARPA)
    assert m.order == 4 and "a" in m and "zzz" not in m and "<unk>" not in m  # (kenlm: index 0 is "not in")
    _run(m, ArpaState, CHAIN, lambda i: m.words[i])
    m2 = ArpaModel(ARPA_NOUNK)
    _run(m2, ArpaState, CHAIN_NOUNK, lambda i: m2.words[i])


def _check_native(lib):
    from pyctcdecode_amd.language_model import NgramModel, NgramState

    m = NgramModel(ARPA)
    assert m.order == 4 and "a" in m and "zzz" not in m and "<unk>" not in m
    _run(m, NgramState, CHAIN, m.word)
    m2 = NgramModel(ARPA_NOUNK)
    _run(m2, NgramState, CHAIN_NOUNK, m2.word)


def test_host_base_score_against_hand_derived_answers(sim_library):  # noqa: F811
    """ctcdec_lm_base_score / ctcdec_lm_start_state (the C ABI behind NgramModel.BaseScore) on the simulator build: the
    host tables are the same code in both libraries."""
    _check_native(sim_library)


# ---- the same arithmetic through a decode: the device's n-gram table, word by word ------------------------------------
LABELS = [" ", "a", "b", "c"]  # + blank appended by the alphabet: V = 5
LN10 = math.log(10.0)
ALPHA, BETA, UNK = _KA["alpha"], _KA["beta"], _KA["unk_score_offset"]
# (spelled text, log10 terms: the words in order, then </s> at the end of the sentence; unknown words)
#   "a b c": Q1 Q2 Q3, then </s> from (a b c): bo(a b c) + bo(b c) + bo(c) + p(</s>) = -0.1875 - 0.5 - 0.125 - 1.5 = -2.3125
#   "a b a": Q1 Q2 Q6, then </s> from the state (a): bo(a) + p(</s>) = -0.25 - 1.5 = -1.75
#   "a bb":  Q1, then the OOV word after <s> a: bo(<s> a) + bo(a) + p(<unk>) = -0.375 - 0.25 - 1.0 = -1.625 (+ the unknown-word
#            offset), then </s> after <unk>: bo(<unk>) + p(</s>) = -1.5
DECODES = [tuple(c) for c in _KA["decodes"]]


def _spell(text):
    """One confident frame per character, two blanks between repeated characters and at the ends."""
    ids = []
    prev = None
    for ch in text:
        k = LABELS.index(ch)
        if k == prev:
            ids.append(4)
        ids.append(k)
        prev = k
    ids = [4] + ids + [4]
    x = np.full((len(ids), 5), -30.0)
    x[np.arange(len(ids)), ids] = 0.0
    return x


def _check_decodes(build_ctcdecoder, to_input):
    dec = build_ctcdecoder(LABELS, ARPA, unigrams=["a", "b", "c"], alpha=ALPHA, beta=BETA, unk_score_offset=UNK)
    for text, terms, n_unk in DECODES:
        beams = dec.decode_beams(to_input(_spell(text)), beam_width=8)
        assert beams[0].text == text, (text, beams[0].text)
        n_words = len(text.split())
        # language_model.py:338-360: alpha * (log10 p [+ unk offset] [+ log10 p(</s>)]) * ln 10 + beta per word
        want = ALPHA * LN10 * (sum(terms) + UNK * n_unk) + BETA * n_words
        got = beams[0].lm_score - beams[0].logit_score
        assert abs(got - want) < 1e-9, (text, got, want)


def test_sim_decode_scores_words_by_the_hand_derived_answers(sim_library, both_beam_kernels):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    _check_decodes(build_ctcdecoder, lambda x: x)


@pytest.mark.gpu
def test_hip_base_score_and_decode_against_hand_derived_answers(both_beam_kernels):
    """The product library: host queries and the device's flat hashed n-gram table (csrc/common.h lm_base_score), through
    decode_beams on cuda:0."""
    import torch

    from pyctcdecode_amd import _binding as B
    from pyctcdecode_amd import build_ctcdecoder

    _check_native(B.get_library())
    _check_decodes(build_ctcdecoder, lambda x: torch.from_numpy(x).cuda())
