"""ORACLE PINNING (this container only): generate tests/golden/* by running the UNMODIFIED
reference (/root/reference, imported through oracle/refshim/) on seeded inputs.

    python oracle/make_golden.py

Outputs (committed):
  tests/golden/cases.json          case descriptions + expected decode_beams outputs
  tests/golden/inputs.npz          the logit matrices (exact dtype fed to the reference)
  tests/golden/bugs_bunny_kenlm.arpa   the reference's toy LM data fixture (19 lines, data)
  tests/golden/known_answers.json  scorer-level known answers
The GPU box has no /root/reference: tests only ever read the files written here.
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
os.makedirs(GOLD, exist_ok=True)
os.chdir("/tmp")

import logging  # noqa: E402
import warnings  # noqa: E402

import numpy as np  # noqa: E402

logging.disable(logging.CRITICAL)
warnings.simplefilter("ignore")

from pyctcdecode import build_ctcdecoder  # noqa: E402  (the reference)
from pyctcdecode.language_model import HotwordScorer  # noqa: E402
import pyctcdecode.tests.test_decoder as rt  # noqa: E402

import synth  # noqa: E402

REF_TOY = "/root/reference/pyctcdecode/tests/sample_data/bugs_bunny_kenlm.arpa"
TOY = os.path.join(GOLD, "bugs_bunny_kenlm.arpa")
with open(REF_TOY) as f:
    toy_text = f.read()
with open(TOY, "w") as f:
    f.write(toy_text)

LM_DIR = os.path.join(GOLD, "_lm")  # regenerated deterministically by tests (git-ignored)
inputs = {}
cases = []


def lm_spec_to_path(spec):
    if spec is None:
        return None, None
    if spec == "toy":
        return TOY, None
    lm = synth.SynthLM(LM_DIR, spec["n_words"], spec["n_sent"], order=spec["order"], seed=spec["seed"],
                       upper=spec.get("upper", False))
    return lm.path, lm


def state_to_json(ref_dec, st):
    if st is None:
        return None
    model = ref_dec._language_model._kenlm_model
    s = st.state
    return {"words": [model.words[i] for i in s.words], "backoff": [float(b) for b in s.backoff]}


def add_case(name, labels, x, lm=None, unigrams=None, build=None, decode=None):
    build = build or {}
    decode = decode or {}
    path, _ = lm_spec_to_path(lm)
    dec = build_ctcdecoder(labels, path, unigrams, **build)
    beams = dec.decode_beams(x, **decode)
    exp = [
        {
            "text": b.text,
            "frames": [[w, int(f[0]), int(f[1])] for w, f in b.text_frames],
            "logit": float(b.logit_score),
            "lm": float(b.lm_score),
            "state": state_to_json(dec, b.last_lm_state),
        }
        for b in beams
    ]
    key = "x%03d" % len(cases)
    inputs[key] = x
    cases.append(
        {"name": name, "labels": labels, "input": key, "lm": lm, "unigrams": unigrams, "build": build,
         "decode": decode, "expected": exp}
    )
    dec.cleanup()
    return beams


# ---- App. E fixtures (reference tests/test_decoder.py:190-223, 245-513, 700-770) -------------
S = rt.SAMPLE_LABELS
TL = rt.TEST_LOGITS
add_case("toy_nolm_16beams", S, TL)
add_case("toy_lm_default", S, TL, lm="toy", unigrams=["bugs", "bunny"])
add_case("toy_lm_autounigrams_prune60", S, TL, lm="toy", decode={"beam_prune_logp": -60.0})
add_case("toy_lm_alpha0", S, TL, lm="toy", unigrams=["bugs", "bunny"], build={"alpha": 0.0})
add_case("toy_lm_alpha1", S, TL, lm="toy", unigrams=["bugs", "bunny"], build={"alpha": 1.0})
add_case("toy_lm_unk0", S, TL, lm="toy", build={"alpha": 1.0, "unk_score_offset": 0.0})
add_case("toy_lm_unk0_prune20", S, TL, lm="toy", unigrams=["bugs", "bunny"],
         build={"unk_score_offset": 0.0}, decode={"beam_prune_logp": -20.0})
add_case("toy_lm_noboundary", S, TL, lm="toy", unigrams=["bugs", "bunny"], build={"lm_score_boundary": False})
add_case("toy_minlogp0_argmax_only", S, TL, decode={"token_min_logp": 0.0})
add_case("toy_history_prune", S, TL, decode={"prune_history": True})
add_case("toy_hotwords_nolm", S, TL, decode={"hotwords": ["bugs"], "hotword_weight": 20.0})
add_case("toy_hotwords_lm", S, TL, lm="toy", unigrams=["bugs", "bunny"], decode={"hotwords": ["bugs"]})
add_case("toy_hotwords_phrase", S, TL, decode={"hotwords": ["bugs bunny", "bun"], "hotword_weight": 10.0})
add_case("toy_T0_nolm", S, TL[:0])
add_case("toy_T0_lm", S, TL[:0], lm="toy", unigrams=["bugs", "bunny"])
add_case("toy_trailing_space_lm", S, np.vstack([TL[:4], rt.TEST_LOGITS[6:7]]), lm="toy",
         unigrams=["bugs", "bunny"], decode={"beam_prune_logp": -60.0})

bpe_labels = ["▁bugs", "▁bun", "ny", ""]
bpe_path = [3, 0, 1, 2, 2, 3]
xb = np.zeros((len(bpe_path), 4))
xb[np.arange(len(bpe_path)), bpe_path] = 1
add_case("bpe_frames", bpe_labels, np.log(np.clip(xb, 1e-15, 1)))

leak_labels = ["<unk>", "▁", "a", "b", "▁a", "▁b"]
xl = np.full((3, 7), 1e-15)
xl[0, 4] = 1.0
xl[1, 1] = 0.5
xl[1, 3] = 0.5
xl[2, 2] = 1.0
add_case("bpe_force_next_break_leak", leak_labels, np.log(xl))

libri = np.asarray(rt.LIBRI_LOGITS)
add_case("libri_char", rt.LIBRI_LABELS, libri)
add_case("libri_char_history", rt.LIBRI_LABELS, libri, decode={"prune_history": True})
spoof = [("▁" if c == " " else c) for c in rt.LIBRI_LABELS]
add_case("libri_spoofed_bpe", spoof, libri)

# ---- seeded random cases ---------------------------------------------------------------------
rng = np.random.default_rng(20260925)
LM_SMALL = {"n_words": 300, "n_sent": 400, "order": 4, "seed": 2}
LM_TRI = {"n_words": 200, "n_sent": 300, "order": 3, "seed": 3}
LM_HF = {"n_words": 300, "n_sent": 400, "order": 4, "seed": 2, "upper": True}
words_small = synth.make_words(300, seed=2)
bpe_vocab_1023 = synth.make_bpe_vocab(words_small, size=1023)
bpe_vocab_127 = synth.make_bpe_vocab(words_small, size=127)


def gen(style, T, V):
    if style == "flat":
        return rng.standard_normal((T, V)).astype(np.float32)
    if style == "peaky":
        x = rng.standard_normal((T, V)).astype(np.float32)
        x[np.arange(T), rng.integers(0, V, size=T)] += np.float32(6.0)
        return x
    if style == "int":
        return rng.integers(-8, 1, size=(T, V)).astype(np.float64)
    e = np.exp(rng.standard_normal((T, V)) * 2)
    return e / e.sum(axis=1, keepdims=True)


def words_case(cfg, utt, T, labels, is_bpe, lm_spec, boost, space="|", blank=None):
    _, lm = lm_spec_to_path(lm_spec)
    if blank is None:
        blank = len(labels)
    return synth.d_words(cfg, utt, T, labels, is_bpe, lm.words, lm.sentences, blank, boost=boost,
                         space_label=space)


n = 0
for vocab_name, labels in [("libri", synth.LIBRI_LABELS), ("hf", synth.HF_W2V2_LABELS),
                           ("bpe127", bpe_vocab_127), ("bpe1023", bpe_vocab_1023)]:
    Vn = len(build_ctcdecoder(labels)._alphabet.labels)
    for style in ["flat", "peaky", "int", "prob"]:
        for lm in [None, LM_HF if vocab_name == "hf" else LM_SMALL]:
            T = int(rng.integers(8, 36)) if Vn < 200 else int(rng.integers(6, 16))
            dkw = {
                "beam_width": int(rng.choice([5, 20, 100])),
                "prune_history": bool(rng.random() < 0.5),
            }
            if rng.random() < 0.3:
                dkw["beam_prune_logp"] = float(rng.choice([-3.0, -30.0]))
            if rng.random() < 0.3:
                dkw["token_min_logp"] = float(rng.choice([-3.0, -8.0]))
            if rng.random() < 0.35:
                src = words_small if vocab_name != "hf" else [w.upper() for w in words_small]
                dkw["hotwords"] = [src[int(i)] for i in rng.integers(0, 300, size=4)] + ["zzqx"]
            x = gen(style, T, Vn).astype(np.float64) if style in ("flat", "peaky") else gen(style, T, Vn)
            add_case("rand_%s_%s_%s_%d" % (vocab_name, style, "lm" if lm else "nolm", n), labels, x, lm=lm,
                     decode=dkw)
            n += 1

# words-like inputs where LM look-ups matter (headline distribution, small sizes)
add_case("words_libri_lm", synth.LIBRI_LABELS,
         words_case(2, 0, 60, synth.LIBRI_LABELS, False, LM_SMALL, 6.0, space=" ").astype(np.float64),
         lm=LM_SMALL, decode={"prune_history": True})
add_case("words_hf_lm_beta1", synth.HF_W2V2_LABELS,
         words_case(3, 0, 60, synth.HF_W2V2_LABELS, False, LM_HF, 6.0, space="|", blank=0).astype(np.float64),
         lm=LM_HF, build={"alpha": 0.5, "beta": 1.0}, decode={"prune_history": True})
add_case("words_bpe1023_lm_hot", bpe_vocab_1023,
         words_case(4, 0, 40, bpe_vocab_1023, True, LM_SMALL, 6.0).astype(np.float64),
         lm=LM_SMALL, decode={"prune_history": True, "hotwords": words_small[:5] + ["qqzzx"]})
add_case("words_bpe1023_lm_beams", bpe_vocab_1023,
         words_case(4, 1, 40, bpe_vocab_1023, True, LM_SMALL, 5.0).astype(np.float64),
         lm=LM_SMALL, decode={"beam_width": 50})
add_case("words_libri_trigram_boost4", synth.LIBRI_LABELS,
         words_case(2, 1, 50, synth.LIBRI_LABELS, False, LM_TRI, 4.0, space=" ").astype(np.float64),
         lm=LM_TRI, decode={"beam_width": 30})

# stateful: second half decoded from the first half's last LM state (tests/test_decoder.py:426-456)
dec = build_ctcdecoder(S, TOY, ["bugs", "bunny"])
first = dec.decode_beams(TL[:5])
state = first[0].last_lm_state
second = dec.decode_beams(TL[7:], lm_start_state=state)
stateful = {
    "first": {"text": first[0].text, "state": state_to_json(dec, state), "lm": float(first[0].lm_score)},
    "second": {"text": second[0].text, "lm": float(second[0].lm_score), "logit": float(second[0].logit_score)},
}
dec.cleanup()

# ---- scorer-level known answers (App. E) ----------------------------------------------------
dec = build_ctcdecoder(S, TOY, ["bugs", "bunny"])
lmod = dec._language_model
st0 = lmod.get_start_state()
s_bugs, st_bugs = lmod.score(st0, "bugs")
known = {
    "lm_score": {
        "<s>->bugs": s_bugs,
        "bugs->bunny(eos)": lmod.score(st_bugs, "bunny", is_last_word=True)[0],
        "bugs->zzz": lmod.score(st_bugs, "zzz")[0],
        "<s>->bunny": lmod.score(st0, "bunny")[0],
    },
    "score_partial": {p: lmod.score_partial_token(p) for p in ["bu", "bux", "buxxxxxx", "b", "bugs", "bunnyx"]},
}
hs = HotwordScorer.build_scorer(["bugs bunny", "bun"], 10.0)
known["hotword_partial"] = {p: hs.score_partial_token(p) for p in ["b", "bu", "bug", "bugs", "bun", "bunn", "x", ""]}
known["hotword_text"] = {t: hs.score(t) for t in ["bugs bunny bun", "bugsy", "", "bun bun", "a bugs"]}
known["stateful"] = stateful
dec.cleanup()

with open(os.path.join(GOLD, "cases.json"), "w") as f:
    json.dump(cases, f, ensure_ascii=False, indent=0)
np.savez_compressed(os.path.join(GOLD, "inputs.npz"), **inputs)
with open(os.path.join(GOLD, "known_answers.json"), "w") as f:
    json.dump(known, f, ensure_ascii=False, indent=1)
print("wrote %d cases, inputs %.1f KB" % (len(cases), os.path.getsize(os.path.join(GOLD, "inputs.npz")) / 1024))
