"""ORACLE PINNING (this container only): golden vectors for the MultiLanguageModel path
(reference language_model.py:455-502) -- the UNMODIFIED reference, imported through oracle/refshim/,
decodes seeded inputs with 2..4 LanguageModels side by side.

    python oracle/make_golden_multi.py

Outputs (committed): tests/golden/cases_multi.json, tests/golden/inputs_multi.npz.
"""
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference", ROOT]
GOLD = os.path.join(ROOT, "tests", "golden")
os.chdir("/tmp")

import logging  # noqa: E402
import warnings  # noqa: E402

import numpy as np  # noqa: E402

logging.disable(logging.CRITICAL)
warnings.simplefilter("ignore")

import kenlm  # noqa: E402  (oracle/refshim stand-in)
from pyctcdecode.alphabet import Alphabet  # noqa: E402  (the reference)
from pyctcdecode.decoder import BeamSearchDecoderCTC  # noqa: E402
from pyctcdecode.language_model import LanguageModel, MultiLanguageModel, load_unigram_set_from_arpa  # noqa: E402
import pyctcdecode.tests.test_decoder as rt  # noqa: E402

import synth  # noqa: E402

TOY = os.path.join(GOLD, "bugs_bunny_kenlm.arpa")
LM_DIR = os.path.join(GOLD, "_lm")
inputs = {}
cases = []


def lm_path(spec):
    if spec == "toy":
        return TOY
    return synth.SynthLM(LM_DIR, spec["n_words"], spec["n_sent"], order=spec["order"], seed=spec["seed"],
                         upper=spec.get("upper", False)).path


def make_lm(member):
    """member: {"lm": spec, "unigrams": list|None|"auto", "build": {alpha, beta, unk_score_offset, score_boundary}}"""
    path = lm_path(member["lm"])
    uni = member.get("unigrams", "auto")
    if uni == "auto":
        uni = sorted(load_unigram_set_from_arpa(path))
    return LanguageModel(kenlm.Model(path), uni, **member.get("build", {}))


def add_case(name, labels, x, members, decode=None):
    decode = decode or {}
    lms = [make_lm(m) for m in members]
    dec = BeamSearchDecoderCTC(Alphabet.build_alphabet(labels), MultiLanguageModel(lms))
    beams = dec.decode_beams(x, **decode)
    exp = []
    for b in beams:
        states = []
        for lm, st in zip(lms, b.last_lm_state.states):
            model = lm._kenlm_model
            states.append({"words": [model.words[i] for i in st.state.words],
                           "backoff": [float(v) for v in st.state.backoff]})
        exp.append({"text": b.text, "frames": [[w, int(f[0]), int(f[1])] for w, f in b.text_frames],
                    "logit": float(b.logit_score), "lm": float(b.lm_score), "states": states})
    key = "m%03d" % len(cases)
    inputs[key] = x
    cases.append({"name": name, "labels": labels, "input": key, "members": members, "decode": decode,
                  "expected": exp})
    dec.cleanup()


S = rt.SAMPLE_LABELS
TL = rt.TEST_LOGITS
TOY_M = {"lm": "toy", "unigrams": ["bugs", "bunny"]}
# the reference's own MultiLM test (tests/test_decoder.py:386-401): twice the same model == the single model
add_case("toy_same_twice", S, TL, [TOY_M, TOY_M])
add_case("toy_alpha_mix", S, TL, [dict(TOY_M, build={"alpha": 1.0}), dict(TOY_M, build={"alpha": 0.0, "beta": 0.5})],
         decode={"beam_prune_logp": -30.0})
add_case("toy_unigram_sets_differ", S, TL,
         [{"lm": "toy", "unigrams": ["bugs"]}, {"lm": "toy", "unigrams": None, "build": {"unk_score_offset": -3.0}},
          {"lm": "toy", "unigrams": ["bunny", "bugs"], "build": {"score_boundary": False}}],
         decode={"beam_prune_logp": -40.0})
add_case("toy_T0", S, TL[:0], [TOY_M, TOY_M])
add_case("toy_hotwords", S, TL, [TOY_M, dict(TOY_M, build={"beta": 0.0})], decode={"hotwords": ["bugs"]})

LM_A = {"n_words": 300, "n_sent": 400, "order": 4, "seed": 2}
LM_B = {"n_words": 200, "n_sent": 300, "order": 3, "seed": 3}
LM_C = {"n_words": 200, "n_sent": 300, "order": 2, "seed": 1}
words_a = synth.make_words(300, seed=2)
bpe = synth.make_bpe_vocab(words_a, size=1023)
lm_a = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
rng = np.random.default_rng(77)


def words_x(cfg, utt, T, labels, is_bpe, boost, space=" "):
    return synth.d_words(cfg, utt, T, labels, is_bpe, lm_a.words, lm_a.sentences, len(labels), boost=boost,
                         space_label=space).astype(np.float64)


add_case("libri_two_orders", synth.LIBRI_LABELS, words_x(2, 0, 60, synth.LIBRI_LABELS, False, 6.0),
         [{"lm": LM_A}, {"lm": LM_B, "build": {"alpha": 0.8, "beta": 1.0}}], decode={"prune_history": True})
add_case("libri_three_models_beams", synth.LIBRI_LABELS, words_x(2, 1, 50, synth.LIBRI_LABELS, False, 4.0),
         [{"lm": LM_B}, {"lm": LM_A, "build": {"unk_score_offset": -5.0}},
          {"lm": LM_C, "unigrams": words_a[:40], "build": {"score_boundary": False}}], decode={"beam_width": 30})
add_case("libri_four_models", synth.LIBRI_LABELS, words_x(2, 2, 40, synth.LIBRI_LABELS, False, 5.0),
         [{"lm": LM_C}, {"lm": LM_B}, {"lm": LM_A}, {"lm": "toy", "unigrams": None}],
         decode={"beam_width": 20, "prune_history": True})
add_case("bpe1023_two_models_hot", bpe, words_x(4, 0, 40, bpe, True, 6.0, space="|"),
         [{"lm": LM_A}, {"lm": LM_B, "build": {"alpha": 0.3}}],
         decode={"prune_history": True, "hotwords": words_a[:5] + ["qqzzx"]})
add_case("bpe1023_flat_two_models", bpe, rng.standard_normal((10, len(bpe) + 1)),
         [{"lm": LM_B}, {"lm": LM_A}], decode={"beam_width": 50})
add_case("libri_flat_two_models", synth.LIBRI_LABELS, rng.standard_normal((25, 29)),
         [{"lm": LM_A}, {"lm": LM_C, "build": {"beta": 0.2}}], decode={"beam_width": 100, "prune_history": True})

# stateful: second half from the first half's MultiLanguageModelState (tests/test_decoder.py:426-456 analogue)
lms = [make_lm(TOY_M), make_lm(dict(TOY_M, build={"alpha": 1.0}))]
dec = BeamSearchDecoderCTC(Alphabet.build_alphabet(S), MultiLanguageModel(lms))
first = dec.decode_beams(TL[:5])
second = dec.decode_beams(TL[7:], lm_start_state=first[0].last_lm_state)
stateful = {"first": first[0].text, "second": {"text": second[0].text, "lm": float(second[0].lm_score),
                                                "logit": float(second[0].logit_score)}}
dec.cleanup()

with open(os.path.join(GOLD, "cases_multi.json"), "w") as f:
    json.dump({"cases": cases, "stateful": stateful}, f, ensure_ascii=False, indent=0)
np.savez_compressed(os.path.join(GOLD, "inputs_multi.npz"), **inputs)
print("wrote %d multi-LM cases" % len(cases))
