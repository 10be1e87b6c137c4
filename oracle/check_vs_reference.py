"""ORACLE PINNING (this container only): differential test of oracle/ctc_oracle.py against the
UNMODIFIED reference imported from /root/reference through the stand-ins in oracle/refshim/.

Run from anywhere:  python oracle/check_vs_reference.py [n_cases] [multi]
It never writes into /root/reference (PYTHONDONTWRITEBYTECODE is forced, cwd is /tmp).
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "refshim"), "/root/reference", ROOT]
os.chdir("/tmp")

import logging  # noqa: E402

import numpy as np  # noqa: E402

logging.disable(logging.CRITICAL)

import pyctcdecode  # noqa: E402  (the reference)
from pyctcdecode import build_ctcdecoder  # noqa: E402

import synth  # noqa: E402
from oracle.ctc_oracle import build_oracle  # noqa: E402

TOY_ARPA = "/root/reference/pyctcdecode/tests/sample_data/bugs_bunny_kenlm.arpa"


def random_case(rng, lm_dir):
    kind = rng.choice(["char", "char_small", "bpe", "bpe_big", "hf"])
    if kind == "char":
        labels = list(synth.LIBRI_LABELS)
    elif kind == "char_small":
        labels = [" ", "b", "g", "n", "s", "u", "y", ""]
    elif kind == "hf":
        labels = list(synth.HF_W2V2_LABELS)
    elif kind == "bpe":
        labels = ["<unk>", "▁", "a", "b", "▁a", "▁b", "▁bu", "gs", "nny", "▁bugs", "n", "y", "s", "g", "u"]
    else:
        words = synth.make_words(300, seed=5)
        labels = synth.make_bpe_vocab(words, size=int(rng.choice([63, 255, 1023])))
    lm_kind = rng.choice(["none", "toy", "synth"])
    arpa = None
    unigrams = None
    if lm_kind == "toy":
        arpa = TOY_ARPA
        if rng.random() < 0.5:
            unigrams = ["bugs", "bunny"]
    elif lm_kind == "synth":
        lm = synth.SynthLM(lm_dir, 200, 300, order=int(rng.choice([2, 3, 4])), seed=int(rng.integers(1, 4)),
                           upper=(kind == "hf"))
        arpa = lm.path
        if rng.random() < 0.3:
            unigrams = lm.words[:50]
    kw = dict(
        alpha=float(rng.choice([0.5, 0.0, 1.0, 0.7])),
        beta=float(rng.choice([1.5, 0.0, 3.0])),
        unk_score_offset=float(rng.choice([-10.0, 0.0, -4.0])),
        lm_score_boundary=bool(rng.random() < 0.7),
    )
    ref = build_ctcdecoder(labels, arpa, unigrams, **kw)
    alpha = ref._alphabet
    orc = build_oracle(alpha.labels, alpha.is_bpe, arpa, unigrams, **kw)
    V = len(alpha.labels)
    T = int(rng.integers(0, 40))
    style = rng.choice(["normal", "peaky", "int", "prob"])
    if style == "normal":
        x = rng.standard_normal((T, V)) * rng.choice([1.0, 1.5, 3.0])
    elif style == "peaky":
        x = rng.standard_normal((T, V))
        if T:
            x[np.arange(T), rng.integers(0, V, size=T)] += 6.0
    elif style == "int":
        x = rng.integers(-8, 1, size=(T, V)).astype(np.float64)
    else:
        e = np.exp(rng.standard_normal((T, V)) * 2)
        x = e / e.sum(axis=1, keepdims=True) if T else e
    hot = None
    if rng.random() < 0.4:
        hot = [str(s) for s in rng.choice(["bugs", "bunny", "bun", "ab", "bugs bunny", "a", "zq"], size=3)]
    dkw = dict(
        beam_width=int(rng.choice([1, 3, 5, 10, 20, 100])),
        beam_prune_logp=float(rng.choice([-3.0, -10.0, -30.0])),
        token_min_logp=float(rng.choice([-5.0, -3.0, -8.0, 0.0])),
        prune_history=bool(rng.random() < 0.5),
        hotwords=hot,
        hotword_weight=float(rng.choice([10.0, 3.0])),
    )
    return ref, orc, x, dkw


def random_multi_case(rng, lm_dir):
    """MultiLanguageModel (language_model.py:455-502): 2-4 member models of different order / unigram sets /
    weights behind the reference decoder vs MultiLMOracle behind the oracle decoder."""
    import kenlm  # the refshim stand-in
    from pyctcdecode.alphabet import Alphabet as RefAlphabet
    from pyctcdecode.decoder import BeamSearchDecoderCTC as RefDecoder
    from pyctcdecode.language_model import LanguageModel as RefLM, MultiLanguageModel as RefMulti

    from oracle.arpa_lm import ArpaModel
    from oracle.ctc_oracle import LMOracle, MultiLMOracle, OracleDecoder, load_unigrams_from_arpa

    kind = rng.choice(["char", "char_small", "bpe_big"])
    if kind == "char":
        labels = list(synth.LIBRI_LABELS)
    elif kind == "char_small":
        labels = [" ", "b", "g", "n", "s", "u", "y", ""]
    else:
        labels = synth.make_bpe_vocab(synth.make_words(300, seed=5), size=int(rng.choice([63, 255])))
    ref_lms, orc_lms = [], []
    for _ in range(int(rng.integers(2, 5))):
        if rng.random() < 0.3:
            path = TOY_ARPA
        else:
            path = synth.SynthLM(lm_dir, 200, 300, order=int(rng.choice([2, 3, 4])), seed=int(rng.integers(1, 4))).path
        r = rng.random()
        unigrams = None if r < 0.2 else (sorted(load_unigrams_from_arpa(path))[: int(rng.integers(1, 60))] if r < 0.5
                                          else sorted(load_unigrams_from_arpa(path)))
        kw = dict(alpha=float(rng.choice([0.5, 0.0, 1.0, 0.7])), beta=float(rng.choice([1.5, 0.0, 3.0])),
                  unk_score_offset=float(rng.choice([-10.0, 0.0, -4.0])), score_boundary=bool(rng.random() < 0.7))
        ref_lms.append(RefLM(kenlm.Model(path), unigrams, **kw))
        orc_lms.append(LMOracle(ArpaModel(path), unigrams, kw["alpha"], kw["beta"], kw["unk_score_offset"],
                                kw["score_boundary"]))
    alpha = RefAlphabet.build_alphabet(labels)
    ref = RefDecoder(alpha, RefMulti(ref_lms))
    orc = OracleDecoder(alpha.labels, alpha.is_bpe, MultiLMOracle(orc_lms))
    V = len(alpha.labels)
    T = int(rng.integers(0, 30))
    x = rng.standard_normal((T, V)) * rng.choice([1.0, 2.0])
    if T and rng.random() < 0.5:
        x[np.arange(T), rng.integers(0, V, size=T)] += 6.0
    hot = None
    if rng.random() < 0.4:
        hot = [str(s) for s in rng.choice(["bugs", "bunny", "bun", "ab", "bugs bunny", "a", "zq"], size=3)]
    dkw = dict(beam_width=int(rng.choice([1, 5, 20, 100])), beam_prune_logp=float(rng.choice([-3.0, -10.0, -30.0])),
               prune_history=bool(rng.random() < 0.5), hotwords=hot)
    return ref, orc, x, dkw


def compare_multi(ref, orc, x, dkw):
    import warnings

    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rb = ref.decode_beams(x, **dkw)
        ob = orc.decode_beams(x, **dkw)
    assert len(rb) == len(ob), (len(rb), len(ob))
    for r, o in zip(rb, ob):
        assert r.text == o[0], (r.text, o[0])
        assert [(w, tuple(f)) for w, f in r.text_frames] == [(w, tuple(f)) for w, f in o[2]]
        assert abs(r.logit_score - o[3]) <= 1e-9 * max(1, abs(o[3])) and abs(r.lm_score - o[4]) <= 1e-9 * max(1, abs(o[4]))
        for rs, os_ in zip(r.last_lm_state.states, o[1]):
            assert tuple(rs.state.words) == tuple(os_.words) and [float(b) for b in rs.state.backoff] == [
                float(b) for b in os_.backoff]


def compare(ref, orc, x, dkw):
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rb = ref.decode_beams(x, **dkw)
            ob = orc.decode_beams(x, **dkw)
    assert len(rb) == len(ob), (len(rb), len(ob))
    for r, o in zip(rb, ob):
        assert r.text == o[0], (r.text, o[0])
        assert [(w, tuple(f)) for w, f in r.text_frames] == [(w, tuple(f)) for w, f in o[2]], (r.text_frames, o[2])
        assert abs(r.logit_score - o[3]) <= 1e-9 * max(1, abs(o[3])), (r.logit_score, o[3])
        assert abs(r.lm_score - o[4]) <= 1e-9 * max(1, abs(o[4])), (r.lm_score, o[4])
        if r.last_lm_state is not None:
            rs = r.last_lm_state.state
            assert tuple(rs.words) == tuple(o[1].words), (rs.words, o[1].words)
            assert [float(b) for b in rs.backoff] == [float(b) for b in o[1].backoff]
        else:
            assert o[1] is None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(12345)
    lm_dir = "/tmp/ctc_oracle_lm"
    if len(sys.argv) > 2 and sys.argv[2] == "multi":
        for i in range(n):
            ref, orc, x, dkw = random_multi_case(rng, lm_dir)
            try:
                compare_multi(ref, orc, x, dkw)
            except AssertionError:
                print("MISMATCH in multi-LM case", i, dkw, x.shape)
                raise
            ref.cleanup()
        print("oracle == reference on %d random MultiLanguageModel cases" % n)
        return
    for i in range(n):
        ref, orc, x, dkw = random_case(rng, lm_dir)
        try:
            compare(ref, orc, x, dkw)
        except AssertionError:
            print("MISMATCH in case", i, dkw, x.shape)
            raise
        ref.cleanup()
    print("oracle == reference on %d random cases" % n)


if __name__ == "__main__":
    main()
