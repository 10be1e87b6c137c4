"""CPU: randomised differential test of the shell + beam_core.h (sequential sim backend) against
the oracle at sizes that exercise chunking (> 512 candidates per frame), pool pruning, 3-way
merges, BPE force_next_break and LM/hot-word fusion.  Continuous inputs => strict order."""
import numpy as np
import pytest

import synth
from oracle.ctc_oracle import build_oracle
from pyctcdecode_amd.alphabet import Alphabet
from tests.golden_util import LM_DIR, check_beams
from tests.sim_util import sim_library  # noqa: F401

pytestmark = pytest.mark.usefixtures("both_beam_kernels")


def _expected(orc, x, kw):
    with np.errstate(all="ignore"):
        out = orc.decode_beams(x, **kw)
    return [{"text": o[0], "frames": [[w, int(a), int(b)] for w, (a, b) in o[2]], "logit": o[3], "lm": o[4]} for o in out]


def _compare(labels, arpa, x, build=None, dkw=None, unigrams=None, what=""):
    from pyctcdecode_amd import build_ctcdecoder

    build = build or {}
    dkw = dkw or {}
    dec = build_ctcdecoder(labels, arpa, unigrams, **build)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe, arpa, unigrams, **build)
    out = dec.decode_beams(x, **dkw)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], _expected(orc, x, dkw),
                tol=1e-9, what=what)
    return dec, orc


LM = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
WORDS = synth.make_words(300, seed=2)
BPE = synth.make_bpe_vocab(WORDS, size=255)


@pytest.mark.parametrize("seed", range(6))
def test_flat_libri_beam100_chunks(seed, sim_library):  # noqa: F811
    x = synth.d_flat(2, seed, 40, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, None, x, dkw={"prune_history": bool(seed % 2)}, what="flat%d" % seed)


@pytest.mark.parametrize("cand", [1024, 2048])
@pytest.mark.parametrize("seed", range(3))
def test_flat_libri_beam100_wide_candidate_chunks(seed, cand, sim_library, monkeypatch):  # noqa: F811
    """The 512-thread variant of the workgroup kernel takes its candidates in chunks of 1024 (beam_core.h: group_cand): a
    frame of ~2 500 candidates then runs through three chunks and one pool compaction (2048: two chunks)."""
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "group")
    monkeypatch.setenv("CTCDEC_SIM_CAND", str(cand))
    x = synth.d_flat(2, 20 + seed, 40, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, None, x, dkw={"prune_history": bool(seed % 2)}, what="flatwide%d_%d" % (cand, seed))
    x = synth.d_flat(3, 30 + seed, 30, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, LM.path, x, dkw={"prune_history": bool(seed % 2), "beam_width": 128},
             what="flatwidelm%d_%d" % (cand, seed))


@pytest.mark.parametrize("seed", range(6))
def test_flat_libri_lm(seed, sim_library):  # noqa: F811
    x = synth.d_flat(3, seed, 30, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, LM.path, x, dkw={"prune_history": bool(seed % 2), "beam_width": 60},
             what="flatlm%d" % seed)


@pytest.mark.parametrize("seed", range(6))
def test_words_bpe_lm_hotwords(seed, sim_library):  # noqa: F811
    x = synth.d_words(4, seed, 50, BPE, True, LM.words, LM.sentences, len(BPE), boost=5.0).astype(np.float64)
    hot = LM.hotwords(4, 2) if seed % 2 else None
    _compare(BPE, LM.path, x, dkw={"prune_history": seed % 3 != 0, "hotwords": hot}, what="bpe%d" % seed)


@pytest.mark.parametrize("seed", range(4))
def test_flat_bpe_small_beam_large_tokens(seed, sim_library):  # noqa: F811
    x = (synth.d_flat(5, seed, 25, len(BPE) + 1) * 1.5).astype(np.float64)
    _compare(BPE, LM.path, x, dkw={"beam_width": 16, "token_min_logp": -8.0, "beam_prune_logp": -30.0},
             what="bpeflat%d" % seed)


def test_decode_and_batch_api(sim_library):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    xs = [synth.d_words(2, u, 20 + 7 * u, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=6.0)
          for u in range(5)] + [np.zeros((0, 29), dtype=np.float32)]
    texts = dec.decode_batch(None, xs)
    assert texts == [orc.decode(x.astype(np.float64)) for x in xs]
    assert dec.decode(xs[1]) == texts[1]
    beams = dec.decode_beams_batch(None, xs[:2], beam_width=20)
    assert [b[0].text for b in beams] == [orc.decode_beams(x.astype(np.float64), beam_width=20)[0][0] for x in xs[:2]]
    assert all(b.last_lm_state is None for bb in beams for b in bb)
    with pytest.raises(ValueError):
        dec.decode(np.zeros((3, 30)))
    with pytest.raises(ValueError):
        dec.decode(np.zeros((3,)))


def _unnormalised_prob_rows(V=200, T=6, seed=3):
    """Rows whose MEAN sum is 1 (=> read as probabilities, decoder.py:760) although single rows hold far
    more than e^5 labels above token_min_logp: exercises the survivor-overflow retry at full width."""
    rng = np.random.default_rng(seed)
    x = rng.random((T, V)) + 0.5
    sums = np.array([0.2, 1.8] * (T // 2))
    x = x / x.sum(axis=1, keepdims=True) * sums[:, None]
    assert abs(x.sum(axis=1).mean() - 1.0) < 1e-12
    return x


def test_probability_rows_overflowing_the_survivor_bound(sim_library):  # noqa: F811
    labels = ["t%d" % i for i in range(199)] + [" "]
    x = _unnormalised_prob_rows(V=201)
    _compare(labels, None, x, dkw={"beam_width": 20}, what="overflow")


def test_three_dimensional_batch_input(sim_library):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    xs = [synth.d_words(2, u, 30, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=6.0) for u in range(4)]
    assert dec.decode_batch(None, np.stack(xs)) == dec.decode_batch(None, xs)
    with pytest.raises(ValueError):
        dec.decode_batch(None, np.zeros((2, 5, 7), dtype=np.float32))


def test_lm_start_state_carry_over(sim_library):  # noqa: F811
    """decode_beams(lm_start_state=prev.last_lm_state) (reference tests/test_decoder.py:426-456)."""
    from pyctcdecode_amd import build_ctcdecoder
    from tests.golden_util import TOY_ARPA, load_cases, load_known

    cases, inputs = load_cases()
    case = {c["name"]: c for c in cases}["toy_lm_default"]
    x = inputs[case["input"]]
    known = load_known()["stateful"]
    dec = build_ctcdecoder(case["labels"], TOY_ARPA, ["bugs", "bunny"])
    first = dec.decode_beams(x[:5])
    assert first[0].text == known["first"]["text"]
    st = first[0].last_lm_state
    lm = dec._language_model._kenlm_model
    assert [lm.word(i) for i in st.state.words] == known["first"]["state"]["words"]
    second = dec.decode_beams(x[7:], lm_start_state=st)
    assert second[0].text == known["second"]["text"]
    assert abs(second[0].lm_score - known["second"]["lm"]) < 1e-12
    assert abs(second[0].logit_score - known["second"]["logit"]) < 1e-12
    # public scorer methods (language_model.py:326-360) against the reference's known answers
    ka = load_known()
    lmod = dec._language_model
    s0 = lmod.get_start_state()
    sc, s1 = lmod.score(s0, "bugs")
    assert sc == ka["lm_score"]["<s>->bugs"]
    assert lmod.score(s1, "bunny", is_last_word=True)[0] == ka["lm_score"]["bugs->bunny(eos)"]
    assert lmod.score(s1, "zzz")[0] == ka["lm_score"]["bugs->zzz"]
    assert lmod.score(s0, "bunny")[0] == ka["lm_score"]["<s>->bunny"]
    for part, val in ka["score_partial"].items():
        assert lmod.score_partial_token(part) == val
    from pyctcdecode_amd.language_model import HotwordScorer

    hs = HotwordScorer.build_scorer(["bugs bunny", "bun"], 10.0)
    for part, val in ka["hotword_partial"].items():
        assert hs.score_partial_token(part) == val
    for text, val in ka["hotword_text"].items():
        assert hs.score(text) == val


def test_non_finite_logits_follow_the_reference(sim_library):  # noqa: F811
    """-inf masked labels decode like any other row (math.isclose(-inf, 1) is False: logits, not
    probabilities); rows that turn into NaN (all -inf, NaN or +inf entries) kill every beam and the
    reference dies on max([]) (decoder.py:545) -- a ValueError here too, never a crash or garbage."""
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(synth.LIBRI_LABELS)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((12, 29)).astype(np.float32)
    masked = x.copy()
    masked[:, 10:20] = -np.inf
    with np.errstate(all="ignore"):
        exp = orc.decode_beams(masked, beam_width=5)  # (float32 logits in their own dtype, as the reference computes them)
    got = dec.decode_beams(masked, beam_width=5)
    assert [g.text for g in got] == [e[0] for e in exp]
    for g, e in zip(got, exp):
        assert abs(g.logit_score - e[3]) < 1e-9
    for poison in ("row", "nan", "pinf", "last"):
        bad = x.copy()
        if poison == "row":
            bad[5, :] = -np.inf
        elif poison == "nan":
            bad[3, 4] = np.nan
        elif poison == "pinf":
            bad[3, 4] = np.inf
        else:
            bad[-1, :] = -np.inf
        with pytest.raises(ValueError), np.errstate(all="ignore"):
            orc.decode_beams(bad.astype(np.float64), beam_width=5)
        with pytest.raises(ValueError):
            dec.decode_beams(bad, beam_width=5)
        with pytest.raises(ValueError):
            dec.decode_batch(None, [x, bad])
    with pytest.raises(ValueError):
        dec.decode_beams(x, beam_prune_logp=1.0)
    assert dec.decode(x) == orc.decode(x.astype(np.float64))  # still healthy afterwards


def test_frame_survivors_in_cpython_set_order(sim_library):  # noqa: F811
    """The prune stage's per-frame label order IS the iteration order of a real CPython set."""
    from pyctcdecode_amd import build_ctcdecoder
    from tests.survivor_util import check_against_cpython

    rng = np.random.default_rng(11)
    # (the near-uniform rows keep every label: 16-18, 64-76 and 256-306 members are the sizes whose union copy
    # outgrows the table the set was built in -- the capacity bound once missed that)
    for V, scale, tmin in [(29, 1.0, -5.0), (29, 3.0, -3.0), (300, 2.0, -5.0), (1024, 1.0, -6.5), (1024, 4.0, -5.0),
                           (17, 0.05, -8.0), (18, 0.05, -8.0), (70, 0.05, -8.0), (76, 0.05, -8.0), (300, 0.05, -8.0)]:
        dec = build_ctcdecoder([chr(0x4E00 + i) for i in range(V - 1)])
        x = (rng.standard_normal((40, V)) * scale).astype(np.float32)
        check_against_cpython(dec, x, tmin, 1e-9)


def test_decode_calls_from_several_host_threads(sim_library):  # noqa: F811
    """Two decoders driven from four host threads at once (ctypes releases the GIL): calls are serialised
    inside the library, every result equals the single-threaded one."""
    import threading

    from pyctcdecode_amd import build_ctcdecoder

    lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
    decs = [build_ctcdecoder(synth.LIBRI_LABELS, lm.path), build_ctcdecoder(synth.LIBRI_LABELS)]
    xs = [synth.d_words(2, u, 40, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=5.0) for u in range(8)]
    want = [[d.decode(x) for x in xs] for d in decs]
    got = [[None] * len(xs) for _ in decs]
    errs = []

    def work(k):
        try:
            for rep in range(3):
                for u in range(k % 2, len(xs), 2):
                    got[k // 2][u] = decs[k // 2].decode(xs[u])
        except Exception as e:  # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs and got == want


def test_random_differential_slice(sim_library):  # noqa: F811
    """A fixed slice of tools/fuzz_sim_vs_oracle.py (random vocabularies / 0-3 language models / hot words /
    decode arguments / input styles / chunkings / batch + stateful entry points) against the oracle."""
    from tools import fuzz_sim_vs_oracle as fuzz

    stats = fuzz.run_many(40, 20260925, tol=1e-9)
    assert sum(stats.values()) == 40 and stats.get("ok", 0) + stats.get("ok+chunked", 0) >= 38, stats


def test_language_models_on_one_ngram_model_keep_their_own_unigram_sets(sim_library):  # noqa: F811
    """The reference builds many LanguageModels with different unigram sets on ONE kenlm.Model
    (tests/test_decoder.py:188-280): building the second must not change how the first scores."""
    from pyctcdecode_amd.decoder import BeamSearchDecoderCTC
    from pyctcdecode_amd.language_model import LanguageModel, NgramModel

    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    x = synth.d_words(2, 3, 40, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=5.0).astype(np.float64)
    model = NgramModel(LM.path)
    half = sorted(LM.words)[: len(LM.words) // 2]
    lm_a = LanguageModel(model, LM.words)
    dec_a = BeamSearchDecoderCTC(alpha, lm_a)
    before = [(o.text, o.lm_score) for o in dec_a.decode_beams(x)]
    lm_b = LanguageModel(model, half)            # same NgramModel, a different unigram set
    dec_b = BeamSearchDecoderCTC(alpha, lm_b)
    got_b = [(o.text, o.lm_score) for o in dec_b.decode_beams(x)]
    after = [(o.text, o.lm_score) for o in dec_a.decode_beams(x)]
    assert after == before
    orc_a = build_oracle(alpha.labels, alpha.is_bpe, LM.path, LM.words)
    orc_b = build_oracle(alpha.labels, alpha.is_bpe, LM.path, half)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in dec_a.decode_beams(x)], _expected(orc_a, x, {}), what="lm_a")
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in dec_b.decode_beams(x)], _expected(orc_b, x, {}), what="lm_b")
    assert got_b != before or half == sorted(LM.words)


def test_kernel_selection_is_reported(sim_library, monkeypatch):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    x = synth.d_flat(2, 0, 12, 29).astype(np.float64)
    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    for kernel, code in (("wave", 1), ("group", 2)):
        monkeypatch.setenv("CTCDEC_BEAM_KERNEL", kernel)
        dec.decode_beams(x)
        assert dec.last_beam_kernel == code
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "wave")
    dec.decode_beams(x, beam_width=200)  # not eligible (beam_width > 128): the workgroup kernel takes it
    assert dec.last_beam_kernel == 2


@pytest.mark.parametrize("seed", range(6))
def test_peaky_posteriors_take_the_single_label_runs(seed, sim_library, monkeypatch):  # noqa: F811
    """Real-posterior-like input (most frames: one survivor, the label every beam already ends in): the wave
    kernel consumes such frames in runs without the full per-frame pipeline -- same beams, frames and scores as
    the oracle, with and without the shortcut."""
    if seed % 2:
        x = synth.d_peaky(8, seed, 120, BPE, True, LM.words, LM.sentences, len(BPE)).astype(np.float64)
        labels = BPE
    else:
        x = synth.d_peaky(8, seed, 120, synth.LIBRI_LABELS, False, LM.words, LM.sentences, len(synth.LIBRI_LABELS),
                          boost=12.0).astype(np.float64)
        labels = synth.LIBRI_LABELS
    hot = LM.hotwords(4, 2) if seed % 3 == 0 else None
    dkw = {"prune_history": seed % 3 != 1, "hotwords": hot, "beam_width": 100 if seed < 4 else 25}
    dec, _ = _compare(labels, LM.path if seed != 4 else None, x, dkw=dkw, what="peaky%d" % seed)
    with_runs = dec.decode_beams(x, **dkw)
    monkeypatch.setenv("CTCDEC_NO_LABEL_RUNS", "1")
    without = dec.decode_beams(x, **dkw)
    assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in with_runs] == \
           [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in without]


def test_label_run_across_look_ahead_windows_to_the_end(sim_library):  # noqa: F811
    """A 200-frame blank stretch (several 64-frame look-ahead windows) up to the last frame, and a held token."""
    labels = synth.LIBRI_LABELS
    V = len(labels) + 1
    head = synth.d_peaky(8, 3, 40, labels, False, LM.words, LM.sentences, len(labels), boost=12.0).astype(np.float64)
    tail = np.full((200, V), -8.0)
    tail[:, V - 1] = 8.0
    held = np.full((70, V), -8.0)
    held[:, 5] = 8.0
    for x, what in ((np.concatenate([head, tail]), "blank-tail"), (np.concatenate([head, held, tail[:3]]), "held")):
        _compare(labels, LM.path, x, dkw={"prune_history": True}, what=what)


def test_decode_batch_texts_with_awkward_labels(sim_library):  # noqa: F811
    """decode_batch takes its texts as one joined buffer and splits it natively; the separator must not be a
    character a label can produce (a newline label moves it on), non-ASCII labels and empty texts must survive."""
    from pyctcdecode_amd import build_ctcdecoder

    for labels in (["\n", "a", "b", "é", " "], ["a", "b", "ü", "\x01", " "], ["\n", "\x01", "\x02", "a", " "]):
        dec = build_ctcdecoder(labels)
        assert dec._texts_sep not in [lab.encode("utf-8") for lab in labels]
        rng = np.random.default_rng(5)
        xs = [rng.standard_normal((t, len(labels) + 1)) * 3 for t in (12, 1, 7, 0, 9)]
        xs[1][:, :] = -20.0
        xs[1][:, len(labels)] = 20.0  # blank only: an empty text in the middle of the batch
        texts = dec.decode_batch(None, xs)
        assert texts == [dec.decode(x) for x in xs]
        assert texts[1] == "" and texts[3] == ""


@pytest.mark.parametrize("kind", ["char", "bpe"])
def test_decode_batch_texts_come_from_the_device(kind, sim_library, both_beam_kernels, monkeypatch):  # noqa: F811
    """decode_batch asks for texts only (ctcdec_params.texts_only): the kernels assemble each best beam's text themselves.
    Same texts as the host replay of the emission lists (CTCDEC_HOST_REPLAY=1), as decode(), and as the other accessors
    of the same result -- ragged batch, an empty utterance, BPE pieces with and without the boundary mark."""
    from pyctcdecode_amd import build_ctcdecoder

    labels = synth.LIBRI_LABELS if kind == "char" else BPE
    dec = build_ctcdecoder(labels, LM.path)
    xs = [synth.d_words(3, u, T, labels, kind == "bpe", LM.words, LM.sentences, len(labels) if kind == "bpe" else 28, boost=6.0)
          for u, T in enumerate([40, 0, 1, 75, 13, 120])]
    xs.append(synth.d_flat(3, 9, 30, xs[0].shape[1]))
    hot = LM.hotwords(3, 1)
    got = dec.decode_batch(None, xs, hotwords=hot)
    monkeypatch.setenv("CTCDEC_HOST_REPLAY", "1")
    assert dec.decode_batch(None, xs, hotwords=hot) == got
    monkeypatch.delenv("CTCDEC_HOST_REPLAY")
    assert got == [dec.decode(x, hotwords=hot) for x in xs]
    assert got[1] == "" and all(t == " ".join(t.split()) for t in got)


def test_output_beams_built_in_c_equal_the_python_ones(sim_library, monkeypatch):  # noqa: F811
    """decode_beams / decode_beams_batch build their OutputBeam lists in one C loop (csrc/pytexts.c: ctcdec_py_output_beams);
    CTCDEC_PY_UNPACK=1 takes the Python loop it replaces: same objects field by field (types included), with and without
    LM states, several LMs, empty utterances, non-ASCII labels; and the instances are the frozen dataclass they claim to be."""
    import dataclasses

    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.decoder import OutputBeam
    from pyctcdecode_amd.language_model import LanguageModel, MultiLanguageModel, NgramModel

    def key(o):
        st = o.last_lm_state
        parts = [] if st is None else (st.states if hasattr(st, "states") else [st])
        assert type(o) is OutputBeam and type(o.text) is str and type(o.logit_score) is float and type(o.lm_score) is float
        assert all(type(f) is tuple and type(f[0]) is str and type(f[1]) is tuple and type(f[1][0]) is int for f in o.text_frames)
        return (o.text, o.text_frames, o.logit_score, o.lm_score, [bytes(p.state.to_c()) for p in parts])

    lm2 = synth.SynthLM(LM_DIR + "_b", 200, 300, order=3, seed=5)
    multi = MultiLanguageModel([LanguageModel(NgramModel(LM.path), LM.words), LanguageModel(NgramModel(lm2.path), lm2.words)])
    from pyctcdecode_amd.alphabet import Alphabet
    from pyctcdecode_amd.decoder import BeamSearchDecoderCTC

    uni = ["\u00e9", "\u4e2d", "a", "b", " "]
    decs = [(build_ctcdecoder(synth.LIBRI_LABELS, LM.path), 29), (build_ctcdecoder(synth.LIBRI_LABELS), 29),
            (BeamSearchDecoderCTC(Alphabet.build_alphabet(synth.LIBRI_LABELS), multi), 29), (build_ctcdecoder(uni), 6)]
    rng = np.random.default_rng(3)
    for dec, V in decs:
        xs = [rng.standard_normal((40, V)) * 3.0, np.zeros((0, V)), rng.standard_normal((7, V)) * 2.0]
        got = {}
        for mode in ("c", "py"):
            if mode == "py":
                monkeypatch.setenv("CTCDEC_PY_UNPACK", "1")
            else:
                monkeypatch.delenv("CTCDEC_PY_UNPACK", raising=False)
            got[mode] = ([[key(o) for o in dec.decode_beams(x, beam_width=20)] for x in xs if len(x)],
                         [[key(o) for o in u] for u in dec.decode_beams_batch(None, xs, beam_width=20)])
        assert got["c"] == got["py"]
        monkeypatch.delenv("CTCDEC_PY_UNPACK", raising=False)
        beam = dec.decode_beams(xs[0], beam_width=20)[0]
        with pytest.raises(dataclasses.FrozenInstanceError):
            beam.text = "x"
        assert dataclasses.replace(beam) == beam and beam.get_mp_safe_beam().text == beam.text


def test_ragged_batch_dispatched_longest_first(sim_library, both_beam_kernels, monkeypatch):  # noqa: F811
    """More utterances than the device holds at once and of different lengths: the beam stage takes them longest first
    (BeamArgs::order; the simulator walks the same order); every result lands in the caller's slot."""
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    lens = [7, 31, 0, 12, 55, 3, 31, 18, 44, 1, 26, 9, 38]
    xs = [synth.d_words(2, u, t, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=5.0) if t
          else np.zeros((0, 29)) for u, t in enumerate(lens)]
    got = dec.decode_beams_batch(None, xs, beam_width=12)
    monkeypatch.setenv("CTCDEC_NO_LPT_ORDER", "1")
    plain = dec.decode_beams_batch(None, xs, beam_width=12)
    single = [dec.decode_beams(x, beam_width=12) for x in xs]
    for g, p, o in zip(got, plain, single):
        key = [(b.text, b.text_frames, b.logit_score, b.lm_score) for b in g]
        assert key == [(b.text, b.text_frames, b.logit_score, b.lm_score) for b in p]
        assert key == [(b.text, b.text_frames, b.logit_score, b.lm_score) for b in o]


def test_node_arenas_outgrown_and_redone(sim_library, both_beam_kernels, monkeypatch, capfd):  # noqa: F811
    """The node arenas are reserved for 16 nodes per frame; flat posteriors over a char vocabulary complete a word for
    every beam in almost every frame (~100 per frame at beam 100): the kernels report the overflow and the beam stage
    is redone with the worst case reserved -- same beams as with the worst case from the start, and as the oracle."""
    from pyctcdecode_amd import build_ctcdecoder

    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    xs = [synth.d_flat(7, u, 70, 29).astype(np.float64) for u in range(3)]
    kw = {"beam_width": 100, "beam_prune_logp": -60.0}  # (a wide threshold keeps the beam full)
    monkeypatch.setenv("CTCDEC_ARENA_TRACE", "1")
    got = build_ctcdecoder(synth.LIBRI_LABELS, LM.path).decode_beams_batch(None, xs, **kw)  # (a fresh decoder: small arenas)
    monkeypatch.setenv("CTCDEC_WORST_CASE_ARENAS", "1")
    full = build_ctcdecoder(synth.LIBRI_LABELS, LM.path).decode_beams_batch(None, xs, **kw)
    for g, f, x in zip(got, full, xs):
        assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in g] == [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in f]
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in g], _expected(orc, x, kw), what="flat")
    assert "node arenas outgrown" in capfd.readouterr().err  # (the first decode did take the redo)
