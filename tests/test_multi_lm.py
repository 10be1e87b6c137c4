"""MultiLanguageModel (reference language_model.py:455-502): the oracle and the sequential-sim build of
the device code against golden vectors the unmodified reference produced (oracle/make_golden_multi.py).
The HIP build of the same path is checked in tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

from oracle.arpa_lm import ArpaModel
from oracle.ctc_oracle import LMOracle, MultiLMOracle, OracleDecoder, load_unigrams_from_arpa
from pyctcdecode_amd.alphabet import Alphabet
from tests.golden_util import GOLD, check_beams, lm_path
from tests.sim_util import sim_library  # noqa: F401

with open(os.path.join(GOLD, "cases_multi.json")) as _f:
    _DATA = json.load(_f)
CASES = _DATA["cases"]
INPUTS = np.load(os.path.join(GOLD, "inputs_multi.npz"))


def member_unigrams(member):
    uni = member.get("unigrams", "auto")
    if uni == "auto":
        return sorted(load_unigrams_from_arpa(lm_path(member["lm"])))
    return uni


def build_oracle_multi(case):
    alpha = Alphabet.build_alphabet(case["labels"])
    lms = []
    for m in case["members"]:
        b = m.get("build", {})
        lms.append(LMOracle(ArpaModel(lm_path(m["lm"])), member_unigrams(m), b.get("alpha", 0.5), b.get("beta", 1.5),
                            b.get("unk_score_offset", -10.0), b.get("score_boundary", True)))
    return OracleDecoder(alpha.labels, alpha.is_bpe, MultiLMOracle(lms))


def build_product_multi(case):
    from pyctcdecode_amd.decoder import BeamSearchDecoderCTC
    from pyctcdecode_amd.language_model import LanguageModel, MultiLanguageModel, NgramModel

    lms = [LanguageModel(NgramModel(lm_path(m["lm"])), member_unigrams(m), **m.get("build", {})) for m in case["members"]]
    return BeamSearchDecoderCTC(Alphabet.build_alphabet(case["labels"]), MultiLanguageModel(lms)), lms


def check_product_case(case):
    dec, lms = build_product_multi(case)
    out = dec.decode_beams(INPUTS[case["input"]], **case["decode"])
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], case["expected"], tol=1e-9,
                what=case["name"])
    for o, e in zip(out, case["expected"]):
        assert len(o.last_lm_state.states) == len(lms)
        for lm, st, es in zip(lms, o.last_lm_state.states, e["states"]):
            assert [lm._kenlm_model.word(i) for i in st.state.words] == es["words"]
            assert [float(np.float32(b)) for b in st.state.backoff] == es["backoff"]
    return dec


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_multi_matches_reference_golden(case):
    orc = build_oracle_multi(case)
    with np.errstate(all="ignore"):
        out = orc.decode_beams(INPUTS[case["input"]], **case["decode"])
    check_beams([(o[0], o[2], o[3], o[4]) for o in out], case["expected"], what=case["name"])
    for o, e in zip(out, case["expected"]):
        for lm, st, es in zip(orc.lm.lms, o[1], e["states"]):
            assert [lm.model.words[i] for i in st.words] == es["words"]
            assert [float(b) for b in st.backoff] == es["backoff"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_sim_multi_matches_reference_golden(case, sim_library):  # noqa: F811
    check_product_case(case)


def test_sim_multi_stateful_and_batch(sim_library):  # noqa: F811
    case = next(c for c in CASES if c["name"] == "toy_same_twice")
    case2 = dict(case, members=[case["members"][0], dict(case["members"][0], build={"alpha": 1.0})])
    dec, _ = build_product_multi(case2)
    x = INPUTS[case["input"]]
    first = dec.decode_beams(x[:5])
    assert first[0].text == _DATA["stateful"]["first"]
    second = dec.decode_beams(x[7:], lm_start_state=first[0].last_lm_state)
    assert second[0].text == _DATA["stateful"]["second"]["text"]
    assert abs(second[0].lm_score - _DATA["stateful"]["second"]["lm"]) < 1e-9
    assert abs(second[0].logit_score - _DATA["stateful"]["second"]["logit"]) < 1e-9
    # batch entry points share the path
    texts = dec.decode_batch(None, [x, x[:5]])
    assert texts[0] == dec.decode(x) and texts[1] == dec.decode(x[:5])
    with pytest.raises(AssertionError):
        dec.decode_beams(x, lm_start_state=first[0].last_lm_state.states[0])


STREAM_CASES = ["toy_alpha_mix", "libri_two_orders", "libri_three_models_beams", "bpe1023_two_models_hot"]


@pytest.mark.parametrize("name", STREAM_CASES)
def test_sim_multi_streaming_equals_whole(name, sim_library):  # noqa: F811
    """partial_decode_beams in chunks (decoder.py:681-728) == decode_beams == the reference's golden beams."""
    check_chunked_case(name, 1e-9)


def check_chunked_case(name, tol):
    from pyctcdecode_amd.language_model import HotwordScorer

    case = next(c for c in CASES if c["name"] == name)
    dec, _ = build_product_multi(case)
    x = INPUTS[case["input"]]
    kw = dict(case["decode"])
    hot = kw.pop("hotwords", None)
    if hot is not None:
        kw["hotword_scorer"] = HotwordScorer.build_scorer(hot, weight=kw.pop("hotword_weight", 10.0))
    T = x.shape[0]
    cuts = [0, T // 3, T // 3 + 1, (2 * T) // 3, T]
    beams, c1, c2 = dec.get_starting_state()
    for a, b in zip(cuts[:-1], cuts[1:]):
        beams = dec.partial_decode_beams(x[a:b], c1, c2, beams, a, is_end=(b == T), **kw)
    got = [(bm.text, list(zip(bm.text.split(), bm.text_frames)), bm.logit_score, bm.lm_score) for bm in beams]
    check_beams(got, case["expected"], tol=tol, what=name + "/chunked")


def test_multi_public_scorer_matches_oracle(sim_library):  # noqa: F811
    """MultiLanguageModel.score / score_partial_token (the public, host-side methods)."""
    case = next(c for c in CASES if c["name"] == "libri_three_models_beams")
    dec, lms = build_product_multi(case)
    orc = build_oracle_multi(case)
    mlm = dec._language_model
    assert mlm.order == orc.lm.order == 4
    st, ost = mlm.get_start_state(), orc.lm.start_state()
    import synth

    words = synth.make_words(300, seed=2)
    for w in [words[3], words[10], "zzzz", words[7]]:
        (s, st), (os_, ost) = mlm.score(st, w), orc.lm.score(ost, w, False)
        assert abs(s - os_) < 1e-12
    for p in ["a", words[5][:2], "qqqqqqqqq", words[8]]:
        assert abs(mlm.score_partial_token(p) - orc.lm.score_partial(p)) < 1e-12


@pytest.mark.parametrize("seed", range(10))
def test_sim_multi_vs_oracle_random(seed, sim_library):  # noqa: F811
    """Random member sets / weights / unigram sets / decode arguments: device logic == oracle (which is pinned
    on the reference by oracle/check_vs_reference.py ... multi)."""
    import synth

    rng = np.random.default_rng(1000 + seed)
    specs = [{"n_words": 300, "n_sent": 400, "order": 4, "seed": 2}, {"n_words": 200, "n_sent": 300, "order": 3, "seed": 3},
             {"n_words": 200, "n_sent": 300, "order": 2, "seed": 1}, "toy"]
    members = []
    for _ in range(int(rng.integers(2, 5))):
        spec = specs[int(rng.integers(0, len(specs)))]
        r = rng.random()
        m = {"lm": spec, "build": {"alpha": float(rng.choice([0.5, 0.0, 1.0])), "beta": float(rng.choice([1.5, 0.0, 3.0])),
                                   "unk_score_offset": float(rng.choice([-10.0, 0.0, -4.0])),
                                   "score_boundary": bool(rng.random() < 0.7)}}
        if r < 0.2:
            m["unigrams"] = None
        elif r < 0.5:
            m["unigrams"] = sorted(load_unigrams_from_arpa(lm_path(spec)))[: int(rng.integers(1, 60))]
        members.append(m)
    bpe = rng.random() < 0.5
    lm_a = synth.SynthLM(os.path.join(GOLD, "_lm"), 300, 400, order=4, seed=2)
    labels = synth.make_bpe_vocab(lm_a.words, size=255) if bpe else list(synth.LIBRI_LABELS)
    case = {"labels": labels, "members": members}
    dec, _ = build_product_multi(case)
    orc = build_oracle_multi(case)
    T = int(rng.integers(1, 50))
    if rng.random() < 0.6:
        x = synth.d_words(2, seed, T, labels, bpe, lm_a.words, lm_a.sentences, len(labels), boost=float(rng.choice([4.0, 6.0])),
                          space_label="|" if bpe else " ").astype(np.float64)
    else:
        x = rng.standard_normal((T, len(labels) + 1))
    kw = {"beam_width": int(rng.choice([1, 5, 30, 100])), "prune_history": bool(rng.random() < 0.5),
          "beam_prune_logp": float(rng.choice([-5.0, -10.0, -30.0]))}
    if rng.random() < 0.4:
        kw["hotwords"] = lm_a.words[:3] + ["zzqx"]
    with np.errstate(all="ignore"):
        exp = orc.decode_beams(x, **kw)
    got = dec.decode_beams(x, **kw)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got],
                [{"text": e[0], "frames": [[w, int(a), int(b)] for w, (a, b) in e[2]], "logit": e[3], "lm": e[4]} for e in exp],
                tol=1e-9, what="multi-rand%d" % seed)
