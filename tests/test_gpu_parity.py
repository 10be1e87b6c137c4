"""GPU (-m gpu): the HIP path through the C ABI against (a) the reference's golden vectors,
(b) the oracle on seeded inputs at sizes it finishes in seconds, (c) size-independent properties
at BASELINE sizes.  Tolerance (ABSOLUTE, see _tol): order and frames exact (near-ties: see golden_util.check_beams),
scores within 1e-9 wherever the device computes in fp64, within the north star's 1e-4 (measured ~1e-6) where it uses the
float32 exponential the reference itself works at."""
import os

import numpy as np
import pytest

import synth
from tests.golden_util import LM_DIR, check_beams, lm_path, load_cases

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("both_beam_kernels", "both_prune_exps")]

CASES, INPUTS = load_cases()
TOL = 1e-9  # absolute (tests/golden_util.check_beams); the device's fp64 scores differ from the oracle's by < 1e-12


def _mode():
    """n: float32 rows in the reference's own float32 arithmetic (the default, round 6); p: round 5's packed polynomial;
    f: everything in fp64 from the exact upcast (tests/conftest.py::both_prune_exps)"""
    return os.environ.get("CTCDEC_PRUNE_EXP", "np")[0]


def _f64(x):
    """What the oracle is fed for a decode of `x`: float32 logits IN THEIR OWN DTYPE when the device computes them the way the
    reference does (the oracle, like the reference, keeps the input dtype through its log-softmax), else the exact upcast."""
    a = np.asarray(x)
    if a.dtype == np.float32 and _mode() == "n":
        return a
    return a.astype(np.float64)


def _tol(x):
    """Bounds for a decode of logits `x` against the oracle (fed _f64(x)). float32 rows: 1e-9 in the default mode -- the device
    restates the reference's float32 log-softmax bit for bit (np_f32.h); under CTCDEC_PRUNE_EXP=pk (up to 4095 labels) the
    packed float32 polynomial: 1e-4 absolute, order exact outside runs closer than 4e-5. 16-bit rows (a multiple of eight labels
    up to 1024) take that polynomial unless CTCDEC_PRUNE_EXP=f64. Everything else is fp64: 1e-9."""
    dt = str(getattr(x, "dtype", "")).replace("torch.", "")
    V = int(x.shape[-1])
    pk = _mode() != "f"
    f32_path = dt == "float32" and V <= 4095 and _mode() == "p"
    # (float16 / bfloat16 rows of a multiple of eight labels: the 64-rows-per-wave kernel widens them and runs the same
    # float32 exponentials; the reference itself computes such rows in float16)
    h_path = dt in ("float16", "bfloat16") and V % 8 == 0 and V <= 1024 and pk
    return {"tol": 1e-4, "tie_tol": 4e-5} if (f32_path or h_path) else {"tol": TOL, "tie_tol": 1e-9}


def _loaded_native():
    from pyctcdecode_amd import _binding as B

    lib = B.get_library()
    assert lib.path.endswith("pyctcdecode_amd/libctcdec.so"), lib.path
    return lib


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_matches_reference_golden(case):
    from pyctcdecode_amd import build_ctcdecoder

    _loaded_native()
    dec = build_ctcdecoder(case["labels"], lm_path(case["lm"]), case["unigrams"], **case["build"])
    out = dec.decode_beams(INPUTS[case["input"]], **case["decode"])
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], case["expected"],
                tol=TOL, what=case["name"])


def _oracle_expected(orc, x, kw):
    with np.errstate(all="ignore"):
        out = orc.decode_beams(x, **kw)
    return [{"text": o[0], "frames": [[w, int(a), int(b)] for w, (a, b) in o[2]], "logit": o[3], "lm": o[4]} for o in out]


@pytest.fixture(scope="module")
def lm():
    return synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)


@pytest.fixture(scope="module")
def bpe(lm):
    return synth.make_bpe_vocab(lm.words, size=1023)


def test_hip_vs_oracle_flat_char_beam100(lm):
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(synth.LIBRI_LABELS)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    xs = [synth.d_flat(2, u, 60, 29) for u in range(4)]
    # device-resident fp32 input, decode_beams_batch
    got = dec.decode_beams_batch(None, [torch.from_numpy(x).cuda() for x in xs], prune_history=True)
    for u, x in enumerate(xs):
        exp = _oracle_expected(orc, _f64(x), {"prune_history": True})
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got[u]], exp, what="flat%d" % u, **_tol(x))


def test_hip_vs_oracle_bpe1024_lm_hotwords(lm, bpe):
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(bpe, lm.path)
    alpha = Alphabet.build_alphabet(bpe)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    hot = lm.hotwords(6, 2)
    xs = [synth.d_words(4, u, 80, bpe, True, lm.words, lm.sentences, len(bpe), boost=6.0) for u in range(3)]
    xs.append(synth.d_flat(4, 7, 40, 1024))
    got = dec.decode_beams_batch(None, [torch.from_numpy(x).cuda() for x in xs], hotwords=hot, prune_history=True)
    for u, x in enumerate(xs):
        exp = _oracle_expected(orc, _f64(x), {"hotwords": hot, "prune_history": True})
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got[u]], exp, what="bpe%d" % u, **_tol(x))
    texts = dec.decode_batch(None, xs, hotwords=hot)
    assert texts == [g[0].text for g in got]


def test_hip_vs_oracle_bpe1025_pieces_plus_blank(lm):
    """The common real shape "1024 BPE pieces + the CTC blank" = 1025 labels: no multiple of four, rows that are only
    4-byte aligned -- the 64-rows-per-wave prune kernel with element-wise loads (round 4), against the oracle."""
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    pieces = synth.make_bpe_vocab(lm.words, size=1024)
    dec = build_ctcdecoder(pieces, lm.path)
    alpha = Alphabet.build_alphabet(pieces)
    assert len(alpha.labels) == 1025
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    hot = lm.hotwords(6, 2)
    xs = [synth.d_words(4, u, 90, pieces, True, lm.words, lm.sentences, len(pieces), boost=6.0) for u in range(3)]
    xs.append(synth.d_flat(4, 9, 40, 1025))
    got = dec.decode_beams_batch(None, [torch.from_numpy(x).cuda() for x in xs], hotwords=hot, prune_history=True)
    for u, x in enumerate(xs):
        exp = _oracle_expected(orc, _f64(x), {"hotwords": hot, "prune_history": True})
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got[u]], exp, what="bpe1025_%d" % u, **_tol(x))
    assert dec.decode_batch(None, xs, hotwords=hot) == [g[0].text for g in got]


@pytest.mark.parametrize("V,scale", [(29, 1.0), (32, 3.0), (2049, 2.5), (2052, 2.5), (3000, 2.5), (4092, 2.5), (4095, 2.5), (4096, 2.5)])
def test_hip_vs_oracle_vocabulary_shapes_of_the_rows64_prune_kernel(V, scale):
    """Round 5 widened the 64-rows-per-wave prune kernel to vocabularies no larger than the survivor bound (character models;
    flat logits make every row dense there: all of them are handed to the per-row kernel) and to 2049 .. 4095 labels (one
    row in flight, 12 / 16 groups of four per lane; 4096 labels stay with the per-row kernel: id 4095 does not fit the 12-bit
    set tables). One decode per shape against the oracle -- survivor order against real CPython sets is
    test_hip_rows64_prune_kernel_shapes."""
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    _loaded_native()
    labels = [chr(0x4E00 + i) for i in range(V - 1)]
    dec = build_ctcdecoder(labels)
    alpha = Alphabet.build_alphabet(labels)
    assert len(alpha.labels) == V
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    rng = np.random.default_rng(V)
    x = (rng.standard_normal((70, V)) * scale).astype(np.float32)
    x[:, 0] += 1.5  # (some blank mass: beams that stay)
    exp = _oracle_expected(orc, _f64(x), {"beam_width": 24})
    got = dec.decode_beams(torch.from_numpy(x).cuda(), beam_width=24)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], exp, what="shape%d" % V, **_tol(x))


def test_hip_ragged_batch_and_edge_cases(lm):
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(synth.LIBRI_LABELS, lm.path)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    xs = [synth.d_words(2, u, T, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=6.0)
          for u, T in enumerate([1, 0, 37, 5, 64, 2])]
    texts = dec.decode_batch(None, xs)
    assert texts == [orc.decode(_f64(x)) for x in xs]
    # beam_width=1 and max beam bucket
    for bw in (1, 200):
        got = dec.decode_beams(xs[4], beam_width=bw)
        exp = _oracle_expected(orc, _f64(xs[4]), {"beam_width": bw})
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], exp, what="bw%d" % bw, **_tol(xs[4]))
    with pytest.raises(NotImplementedError):
        dec.decode_beams(xs[4], beam_width=300)
    with pytest.raises(ValueError):
        dec.decode(np.zeros((4, 7), dtype=np.float32))


def test_hip_full_size_properties(lm, bpe):
    """BASELINE-size frames (T=1000, V=1024, beam 100, 4-gram, hot words): the oracle is too slow for
    a batch, so check size-independent properties: batch == single, device == host input,
    permutation invariance, determinism, frame monotonicity, score identity."""
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(bpe, lm.path)
    hot = lm.hotwords(6, 2)
    xs = [synth.d_words(4, u, 1000, bpe, True, lm.words, lm.sentences, len(bpe), boost=6.0) for u in range(6)]
    dev = [torch.from_numpy(x).cuda() for x in xs]
    a = dec.decode_beams_batch(None, dev, hotwords=hot, prune_history=True)
    b = dec.decode_beams_batch(None, xs, hotwords=hot, prune_history=True)  # host input path
    c = dec.decode_beams_batch(None, dev[::-1], hotwords=hot, prune_history=True)[::-1]
    single = [dec.decode_beams(x, hotwords=hot, prune_history=True) for x in dev[:2]]
    for u in range(6):
        ta = [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in a[u]]
        assert ta == [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in b[u]]
        assert ta == [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in c[u]]
        if u < 2:
            assert ta == [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in single[u]]
        scores = [o.lm_score for o in a[u]]
        assert scores == sorted(scores, reverse=True) and len(scores) >= 1
        assert scores[0] - scores[-1] <= 10.0 + 1e-9
        for o in a[u]:
            assert len(o.text.split()) == len(o.text_frames)
            ends = [f[1][1] for f in o.text_frames]
            starts = [f[1][0] for f in o.text_frames]
            assert all(0 <= s < e <= 1000 for s, e in zip(starts, ends))
            assert all(starts[k + 1] >= ends[k] - 0 for k in range(len(starts) - 1))


def test_hip_oracle_one_full_length_utterance(lm, bpe):
    """One full-size utterance (T=1000, V=1024, beam 100, 4-gram + hot words) against the oracle."""
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(bpe, lm.path)
    alpha = Alphabet.build_alphabet(bpe)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    hot = lm.hotwords(6, 2)
    x = synth.d_words(4, 11, 1000, bpe, True, lm.words, lm.sentences, len(bpe), boost=6.0)
    got = dec.decode_beams(x, hotwords=hot, prune_history=True)
    exp = _oracle_expected(orc, _f64(x), {"hotwords": hot, "prune_history": True})
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], exp, what="full", **_tol(x))


def test_hip_config2_full_size_char_nolm_stress():
    """BASELINE config 2 at full size (29-char alphabet, no LM, beam 100, batch 256 x T=1000) on the
    stress distribution D_flat (~2 500 candidates per frame: chunked merge + pool pruning + bitonic
    path). The oracle needs ~17 s per such utterance, so: size-independent properties at full size, and
    one T=120 prefix-free utterance against the oracle."""
    import time

    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(synth.LIBRI_LABELS)
    xs = [synth.d_flat(2, u, 1000, 29) for u in range(256)]
    dev = [torch.from_numpy(x).cuda() for x in xs]
    dec.decode_batch(None, dev[:4])
    t0 = time.perf_counter()
    texts = dec.decode_batch(None, dev)
    dt = time.perf_counter() - t0
    print("config2 D_flat: %.1f ms for 256x1000 frames (%.2f M frames/s)" % (1e3 * dt, 256e3 / dt / 1e6))
    assert len(texts) == 256 and all(isinstance(t, str) and len(t) > 0 for t in texts)
    assert texts == dec.decode_batch(None, dev[::-1])[::-1]          # permutation invariance
    assert texts[:3] == [dec.decode(x) for x in xs[:3]]               # batch == single, host == device
    beams = dec.decode_beams(dev[0], prune_history=True)
    assert beams[0].text == texts[0]
    assert [b.lm_score for b in beams] == sorted((b.lm_score for b in beams), reverse=True)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    x = synth.d_flat(2, 999, 120, 29)
    exp = _oracle_expected(orc, _f64(x), {})
    got = dec.decode_beams(x)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], exp, what="cfg2", **_tol(x))


def test_hip_dense_small_vocabulary_calls_skip_the_rows64_prune_kernel(monkeypatch):
    """Round 6: a small vocabulary fed flat logits overflows the 64-rows-per-wave prune kernel in nearly every row (every label
    survives); the caller notices and sends the decoder's next calls straight to one wave per row (PruneArgs::dense_hint). Same
    survivors either way: the first call (fast kernel + listed rows), the following calls (hinted), a call under
    CTCDEC_PRUNE_KERNEL=row and one of real-posterior-like rows in between agree beam for beam with what they gave before."""
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    if _mode() != "n":
        pytest.skip("the hint is honoured only where both prune kernels give the same bits (the default float32 arithmetic)")
    dec = build_ctcdecoder(synth.LIBRI_LABELS)
    flat = [torch.from_numpy(synth.d_flat(2, 300 + u, 160, 29)).cuda() for u in range(8)]  # 1 280 rows, ~25 survivors each
    tup = lambda out: [[(b.text, tuple(b.text_frames), b.logit_score, b.lm_score) for b in beams] for beams in out]  # noqa: E731
    first = tup(dec.decode_beams_batch(None, flat, beam_width=32))
    hinted = tup(dec.decode_beams_batch(None, flat, beam_width=32))
    assert hinted == first
    monkeypatch.setenv("CTCDEC_PRUNE_KERNEL", "row")
    rows = tup(dec.decode_beams_batch(None, flat, beam_width=32))
    monkeypatch.delenv("CTCDEC_PRUNE_KERNEL")
    assert rows == first
    fresh = build_ctcdecoder(synth.LIBRI_LABELS)  # (no hint yet)
    rng = np.random.default_rng(5)
    peaky = []
    for u in range(8):  # rows with one or two survivors: what the fast kernel is for
        x = rng.normal(size=(200, 29)).astype(np.float32)
        x[np.arange(200), rng.integers(0, 29, size=200)] += 14.0
        peaky.append(torch.from_numpy(x).cuda())
    assert tup(fresh.decode_beams_batch(None, peaky, beam_width=32)) == tup(dec.decode_beams_batch(None, peaky, beam_width=32))
    assert tup(dec.decode_beams_batch(None, flat, beam_width=32)) == first


def test_hip_config3_hf_vocab_lm_full_length():
    """BASELINE config 3 shape (HF Wav2Vec2 char vocab V=32, 4-gram, alpha 0.5 beta 1.0, beam 100):
    a ragged batch incl. one T=1000 utterance against the oracle."""
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    lm_u = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2, upper=True)
    kw = {"alpha": 0.5, "beta": 1.0}
    dec = build_ctcdecoder(synth.HF_W2V2_LABELS, lm_u.path, **kw)
    alpha = Alphabet.build_alphabet(synth.HF_W2V2_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm_u.path, None, **kw)
    xs = [synth.d_words(3, u, T, synth.HF_W2V2_LABELS, False, lm_u.words, lm_u.sentences, 0, boost=6.0,
                        space_label="|") for u, T in enumerate([1000, 130, 77])]
    got = dec.decode_batch(None, xs)
    assert got == [orc.decode(_f64(x)) for x in xs]
    beams = dec.decode_beams(xs[1])
    exp = _oracle_expected(orc, _f64(xs[1]), {})
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in beams], exp, what="cfg3", **_tol(xs[1]))


def test_hip_streaming_matches_reference_scenarios():
    """partial_decode_beams on the device: chunked == unchunked == decode_beams on the reference's
    fixture (tests/test_decoder.py:515-584), with and without LM."""
    from pyctcdecode_amd import build_ctcdecoder
    from tests.golden_util import TOY_ARPA

    by_name = {c["name"]: c for c in CASES}
    labels = by_name["toy_nolm_16beams"]["labels"]
    x = INPUTS[by_name["toy_nolm_16beams"]["input"]]
    for lm_path_ in (None, TOY_ARPA):
        dec = build_ctcdecoder(labels, lm_path_)
        beams, c1, c2 = dec.get_starting_state()
        whole = dec.partial_decode_beams(x, c1, c2, beams, 0, is_end=True)
        beams, c1, c2 = dec.get_starting_state()
        beams = dec.partial_decode_beams(x[:3], c1, c2, beams, 0)
        beams = dec.partial_decode_beams(x[3:8], c1, c2, beams, 3)
        parts = dec.partial_decode_beams(x[8:], c1, c2, beams, 8, is_end=True)
        full = dec.decode_beams(x)
        assert len(whole) == len(parts) == len(full)
        assert parts[0].text == ("bunny bunny" if lm_path_ is None else "bugs bunny")
        assert sorted((p.text, tuple(p.text_frames)) for p in parts) == sorted(
            (f.text, tuple(t[1] for t in f.text_frames)) for f in full)
        assert abs(parts[0].logit_score - full[0].logit_score) < 1e-9


def test_hip_config5_streaming_64_streams(lm, bpe):
    """BASELINE config 5: V=1024, 4-gram LM, beam=200, 64 concurrent streams x 50-frame chunks, one
    launch per chunk. Streams 0-1 are checked against the oracle's chunked decode; every stream must
    equal the unchunked device decode."""
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(bpe, lm.path)
    n_streams, n_chunks, chunk = 64, 4, 50
    xs = [synth.d_words(5, u, n_chunks * chunk, bpe, True, lm.words, lm.sentences, len(bpe), boost=6.0)
          for u in range(n_streams)]
    states = [dec.get_starting_state() for _ in range(n_streams)]
    beams = [s[0] for s in states]
    for k in range(n_chunks):
        dev = [torch.from_numpy(x[k * chunk:(k + 1) * chunk]).cuda() for x in xs]
        beams = dec.partial_decode_beams_batch(dev, [s[1] for s in states], [s[2] for s in states], beams,
                                               [k * chunk] * n_streams, beam_width=200,
                                               is_end=(k == n_chunks - 1))
    whole = dec.decode_beams_batch(None, xs, beam_width=200)
    for u in range(n_streams):
        assert [b.text for b in beams[u]] == [w.text for w in whole[u]]
        assert [b.text_frames for b in beams[u]] == [[f[1] for f in w.text_frames] for w in whole[u]]
        assert all(abs(b.lm_score - w.lm_score) < 1e-9 for b, w in zip(beams[u], whole[u]))
    alpha = Alphabet.build_alphabet(bpe)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    for u in range(2):
        st = orc.get_starting_state()
        exp = None
        for k in range(n_chunks):
            exp = orc.partial_decode_beams(_f64(xs[u][k * chunk:(k + 1) * chunk]), st, k * chunk,
                                           beam_width=200, is_end=(k == n_chunks - 1))
        assert [b.text for b in beams[u]] == [e.text for e in exp]
        assert [[tuple(f) for f in b.text_frames] for b in beams[u]] == [[tuple(f) for f in e.tframes] for e in exp]
        assert all(abs(b.lm_score - e.lm) < _tol(xs[u])["tol"] for b, e in zip(beams[u], exp))


def test_hip_half_precision_logits_in_place(lm, bpe):
    """fp16 / bf16 device logits are read in place (exact widening); the result must equal decoding the
    same values handed over as fp32 (beams exactly, scores to 1e-6) -- and the oracle's decode of those widened values
    (the comparison with itself alone would only show that the widening is exact)."""
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    dec = build_ctcdecoder(bpe, lm.path)
    alpha = Alphabet.build_alphabet(bpe)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    x = synth.d_words(4, 21, 300, bpe, True, lm.words, lm.sentences, len(bpe), boost=6.0)
    for dt in (torch.float16, torch.bfloat16):
        xh = torch.from_numpy(x).cuda().to(dt)
        a = dec.decode_beams(xh, prune_history=True)
        b = dec.decode_beams(xh.to(torch.float32), prune_history=True)
        # (same beams; both go through the 64-rows-per-wave kernel by default -- the half types widened on the fly -- and
        # through the per-row kernels under CTCDEC_PRUNE_EXP=f64, where the generic one (half types) and the
        # register-resident one (float32) sum the row in different orders: ~1e-11)
        assert [(o.text, o.text_frames) for o in a] == [(o.text, o.text_frames) for o in b]
        # (the float32 copy is computed in the reference's float32 arithmetic by default, the 16-bit rows never are: the
        #  float32 bound between the two unless everything is fp64)
        bound = _tol(xh)["tol"]
        for o, q in zip(a, b):
            assert abs(o.logit_score - q.logit_score) <= bound
            assert abs(o.lm_score - q.lm_score) <= bound
        wide = xh.to(torch.float32).cpu().numpy()
        with np.errstate(all="ignore"):
            exp = orc.decode_beams(wide.astype(np.float64), prune_history=True)
        expd = [{"text": e[0], "frames": [[w, int(f0), int(f1)] for w, (f0, f1) in e[2]], "logit": e[3], "lm": e[4]} for e in exp]
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in a], expd, what="half %s" % dt, **_tol(xh))


def test_hip_probability_rows_overflowing_the_survivor_bound():
    """Rows read as probabilities (mean row sum 1) whose single rows exceed the e^5 survivor bound: the
    prune stage reports the overflow and both stages are redone at full width."""
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet
    from tests.test_sim_vs_oracle import _unnormalised_prob_rows

    labels = ["t%d" % i for i in range(199)] + [" "]
    x = _unnormalised_prob_rows(V=201)
    dec = build_ctcdecoder(labels)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    got = dec.decode_beams(x, beam_width=20)
    exp = _oracle_expected(orc, x, {"beam_width": 20})
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], exp, tol=TOL, what="overflow")


# ---- MultiLanguageModel (language_model.py:455-502) on the HIP path ---------------------------------
from tests import test_multi_lm as _multi  # noqa: E402  (case loaders only; its own tests are not gpu-marked)


@pytest.mark.parametrize("case", _multi.CASES, ids=[c["name"] for c in _multi.CASES])
def test_hip_multi_lm_matches_reference_golden(case):
    _loaded_native()
    dec, lms = _multi.build_product_multi(case)
    out = dec.decode_beams(_multi.INPUTS[case["input"]], **case["decode"])
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], case["expected"], tol=TOL,
                what=case["name"])
    for o, e in zip(out, case["expected"]):
        for m, st, es in zip(lms, o.last_lm_state.states, e["states"]):
            assert [m._kenlm_model.word(i) for i in st.state.words] == es["words"]


def test_hip_multi_lm_vs_oracle_batch(lm, bpe):
    """Two models of different order over the BPE-1024 vocabulary, a batch of word-like utterances."""
    import torch

    case = {"labels": bpe, "members": [{"lm": {"n_words": 300, "n_sent": 400, "order": 4, "seed": 2}},
                                       {"lm": {"n_words": 200, "n_sent": 300, "order": 3, "seed": 3},
                                        "build": {"alpha": 0.8, "beta": 1.0}}]}
    dec, _ = _multi.build_product_multi(case)
    orc = _multi.build_oracle_multi(case)
    xs = [synth.d_words(4, u, 120, bpe, True, lm.words, lm.sentences, len(bpe), boost=6.0, space_label="|")
          for u in range(6)]
    hot = lm.hotwords(4, 1)
    got = dec.decode_beams_batch(None, [torch.from_numpy(x).cuda() for x in xs], prune_history=True, hotwords=hot)
    for u, x in enumerate(xs):
        exp = _oracle_expected(orc, _f64(x), {"prune_history": True, "hotwords": hot})
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got[u]], exp, what="multi%d" % u, **_tol(x))
    texts = dec.decode_batch(None, torch.from_numpy(np.stack(xs)).cuda(), hotwords=hot)
    assert texts == [g[0].text for g in got]


@pytest.mark.parametrize("name", _multi.STREAM_CASES)
def test_hip_multi_lm_streaming_equals_reference_golden(name):
    _loaded_native()
    _multi.check_chunked_case(name, TOL)


def test_hip_non_finite_logits_follow_the_reference():
    """-inf masked labels are ordinary logits; NaN rows end in the reference's ValueError, not in a fault."""
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    _loaded_native()
    dec = build_ctcdecoder(synth.LIBRI_LABELS)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((40, 29)).astype(np.float32)
    masked = x.copy()
    masked[:, 10:20] = -np.inf
    with np.errstate(all="ignore"):
        exp = _oracle_expected(orc, _f64(masked), {"beam_width": 20})
    got = dec.decode_beams(torch.from_numpy(masked).cuda(), beam_width=20)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], exp, what="masked", **_tol(masked))
    big = rng.standard_normal((30, 1024)).astype(np.float32)  # the register-resident prune kernel
    dec_big = build_ctcdecoder([chr(0x4E00 + i) for i in range(1023)])
    for poison in ("row", "nan", "pinf", "last"):
        for base, d in ((x, dec), (big, dec_big)):
            bad = base.copy()
            if poison == "row":
                bad[5, :] = -np.inf
            elif poison == "nan":
                bad[3, 4] = np.nan
            elif poison == "pinf":
                bad[3, 4] = np.inf
            else:
                bad[-1, :] = -np.inf
            with pytest.raises(ValueError):
                d.decode_beams(bad, beam_width=5)
            with pytest.raises(ValueError):
                d.decode_batch(None, [base, bad])
    assert dec.decode(x) == orc.decode(_f64(x))
    assert len(dec_big.decode(big)) > 0


def test_hip_frame_survivors_in_cpython_set_order():
    """The prune kernels' per-frame label order against a REAL CPython set (the wave-distributed table for
    up to 18 survivors, the lane-0 LDS walk above that; register-resident and generic kernels; fp32/fp64/fp16)."""
    from pyctcdecode_amd import build_ctcdecoder
    from tests.survivor_util import check_against_cpython

    _loaded_native()
    rng = np.random.default_rng(11)
    border = frames = 0
    for V, scale, tmin, dt in [(29, 1.0, -5.0, np.float32), (29, 3.0, -3.0, np.float64), (300, 2.0, -5.0, np.float32),
                               (1024, 1.0, -6.5, np.float32), (1024, 1.0, -7.5, np.float32), (1024, 4.0, -5.0, np.float32),
                               (1024, 2.0, -5.0, np.float16), (5000, 3.0, -6.0, np.float32), (29, 0.3, -3.2, np.float32),
                               # near-uniform rows keep every label: member counts whose union copy outgrows the table
                               # the set was built in (16-18 in the wave-resident table, 64-76 / 256-306 in LDS)
                               (17, 0.05, -8.0, np.float32), (18, 0.05, -8.0, np.float64), (70, 0.05, -8.0, np.float32),
                               (76, 0.05, -8.0, np.float32), (300, 0.05, -8.0, np.float32)]:
        dec = build_ctcdecoder([chr(0x4E00 + i) for i in range(V - 1)])
        x = (rng.standard_normal((300, V)) * scale).astype(dt)
        border += check_against_cpython(dec, x, tmin, _tol(x)["tol"])
        frames += 300
    assert border < frames // 20


def test_hip_rows64_prune_kernel_shapes(monkeypatch):
    """frame_prune_fast (64 rows per wave, one row per lane for the set order) where its shape matters: partial last
    batches, label counts that do not fill the 256-label chunks, rows it hands to the per-row kernel (non-finite rows,
    more than 16 candidates, 15 < survivors), and the same rows through the per-row kernel (CTCDEC_PRUNE_KERNEL=row): equal
    lists, equal to a real CPython set."""
    from pyctcdecode_amd import build_ctcdecoder
    from tests.survivor_util import check_against_cpython, survivors

    _loaded_native()
    rng = np.random.default_rng(23)
    frames = border = 0
    for V, T, scale, tmin in [(1024, 1, 2.0, -5.0), (1024, 63, 2.0, -5.0), (1024, 64, 3.0, -4.0), (1024, 65, 2.0, -5.0),
                              (1024, 200, 1.3, -5.5), (1020, 130, 2.0, -5.0), (772, 70, 2.0, -5.0), (512, 129, 2.0, -4.5),
                              (260, 67, 2.0, -4.0), (256, 64, 1.5, -4.0), (32, 500, 1.0, -3.0), (64, 100, 0.7, -3.9),
                              (8, 77, 1.0, -1.5), (4, 5, 1.0, -1.0),
                              # round 4: label counts that are no multiple of four (each lane fetches its labels one by one:
                              # "1024 pieces + blank") and more than 1024 labels (five to eight groups of four per lane)
                              (1025, 130, 2.0, -5.0), (1027, 70, 2.0, -5.0), (1021, 64, 2.0, -5.0), (29, 200, 1.0, -3.0),
                              (30, 65, 1.0, -3.0), (31, 64, 1.2, -3.0), (5, 66, 1.0, -1.5), (1280, 65, 2.0, -5.0),
                              (1540, 64, 2.0, -5.5), (2044, 66, 2.0, -5.5), (2046, 64, 2.2, -6.0), (2045, 67, 2.0, -5.5), (2048, 64, 2.0, -5.5), (2047, 65, 2.0, -5.5),
                              # round 5: 2049 .. 4095 labels (12 / 16 groups of four per lane, one row in flight), 4096 (per-row
                              # kernel: id 4095 does not fit the set tables), and small vocabularies whose rows are dense (every label
                              # survives: every row goes to the per-row kernel) or sparse
                              (2049, 66, 2.0, -5.5), (2052, 64, 2.0, -5.5), (2560, 65, 2.2, -5.5), (3000, 70, 2.0, -5.5), (3072, 64, 2.0, -6.0),
                              (4092, 65, 2.0, -5.5), (4093, 64, 2.0, -5.5), (4095, 66, 2.0, -5.5), (4096, 64, 2.0, -5.5),
                              (29, 200, 1.0, -6.0), (40, 130, 0.5, -5.0), (29, 130, 4.0, -5.0), (64, 65, 3.0, -5.0), (150, 64, 2.0, -5.0)]:
        dec = build_ctcdecoder([chr(0x4E00 + i) for i in range(V - 1)])
        x = (rng.standard_normal((T, V)) * scale).astype(np.float32)
        if T >= 64:
            x[5, :] = -np.inf           # an all-masked row
            x[9, 3] = np.inf
            x[11, V // 2] = np.nan
            x[17, : V // 2] = -np.inf   # half the labels masked: a clean row
            x[20, :] = 0.25             # every label ties for the maximum
            x[21, :] = 0.0
            x[21, [V - 1, V // 3]] = 9.0  # two maxima: the first one is the argmax
        good = np.ones(T, bool)
        if T >= 64:
            good[[5, 9, 11]] = False
        if V % 8 == 0 and T in (65, 129, 500, 64):  # the same rows as float16: the widening variant of the kernel
            x = x.astype(np.float16)
            if T >= 64:
                x[21, :] = 0.0
                x[21, [V - 1, V // 3]] = 9.0
        monkeypatch.delenv("CTCDEC_PRUNE_KERNEL", raising=False)
        fast = survivors(dec, x, tmin)
        monkeypatch.setenv("CTCDEC_PRUNE_KERNEL", "row")
        per_row = survivors(dec, x, tmin)
        monkeypatch.delenv("CTCDEC_PRUNE_KERNEL")
        for t in range(T):
            assert fast[t][0] == per_row[t][0], (V, T, t)
            if good[t]:
                np.testing.assert_allclose(fast[t][1], per_row[t][1], rtol=0, atol=2e-6)
        border += check_against_cpython(dec, x[good], tmin, 1e-4)
        frames += int(good.sum())
    assert border < frames // 20


def test_hip_random_differential_slice():
    """The same random differential hunt as tests/test_sim_vs_oracle.py, against the HIP build."""
    from tools import fuzz_sim_vs_oracle as fuzz

    _loaded_native()
    stats = fuzz.run_many(60, 20260926, tol=TOL, tol_f32=None if os.environ.get("CTCDEC_PRUNE_EXP") == "f64" else 1e-4)
    assert sum(stats.values()) == 60 and stats.get("ok", 0) + stats.get("ok+chunked", 0) >= 57, stats


@pytest.mark.parametrize("threads", ["256", "512"])
def test_hip_workgroup_kernel_with_four_and_eight_waves(lm, bpe, threads, monkeypatch):
    """The workgroup kernel runs an utterance on eight waves when every CU holds at most one utterance, on four otherwise
    (CTCDEC_GROUP_THREADS forces one): the same beams either way, and the oracle's."""
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    _loaded_native()
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "group")
    monkeypatch.setenv("CTCDEC_GROUP_THREADS", threads)
    for labels, is_bpe, bw, kw in ((synth.LIBRI_LABELS, False, 100, {}), (bpe, True, 200, {"hotwords": lm.hotwords(5, 3)}),
                                   (bpe, True, 30, {})):
        dec = build_ctcdecoder(labels, lm.path)
        alpha = Alphabet.build_alphabet(labels)
        orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
        xs = [synth.d_words(6, u, 90, labels, is_bpe, lm.words, lm.sentences, len(labels), boost=4.0)
              for u in range(3)]
        xs.append(synth.d_flat(6, 9, 60, len(labels) + 1))
        got = dec.decode_beams_batch(None, xs, beam_width=bw, **kw)
        for x, g in zip(xs, got):
            with np.errstate(all="ignore"):
                exp = orc.decode_beams(_f64(x), beam_width=bw, **kw)
            expd = [{"text": e[0], "frames": [[w, int(a), int(b)] for w, (a, b) in e[2]], "logit": e[3], "lm": e[4]} for e in exp]
            check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in g], expd, what="threads " + threads, **_tol(x))


def test_hip_ragged_batch_is_dispatched_longest_first(lm, monkeypatch):
    """A ragged batch of more utterances than the device holds at once hands the workgroups out longest utterance first
    (BeamArgs::order); results land where the caller's order says, equal to the plain dispatch and to the oracle."""
    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    _loaded_native()
    rng = np.random.default_rng(77)
    dec = build_ctcdecoder(synth.LIBRI_LABELS, lm.path)
    lens = [int(t) for t in rng.integers(0, 60, size=700)]
    lens[3] = 0
    lens[10] = 150
    xs = [synth.d_words(9, u, t, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=5.0).astype(np.float32)
          if t else np.zeros((0, 29), np.float32) for u, t in enumerate(lens)]
    got = dec.decode_batch(None, xs, beam_width=16)
    beams = dec.decode_beams_batch(None, xs, beam_width=16)
    monkeypatch.setenv("CTCDEC_NO_LPT_ORDER", "1")
    plain = dec.decode_batch(None, xs, beam_width=16)
    assert got == plain and [b[0].text for b in beams] == got
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    for u in (0, 3, 10, 11, 350, 699):
        assert got[u] == orc.decode(_f64(xs[u]), beam_width=16), u


def test_hip_device_binding_and_kernel_choice(lm, monkeypatch):
    """One process drives one GPU (LOCAL_RANK / CTCDEC_DEVICE), logits on another device are refused, and the two
    beam kernels are chosen by batch size unless CTCDEC_BEAM_KERNEL says otherwise."""
    import torch

    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.language_model import _default_device

    lib = _loaded_native()
    dec = build_ctcdecoder(synth.LIBRI_LABELS, lm.path)
    assert lib.dll.ctcdec_device() == _default_device() == dec._device
    x = torch.from_numpy(synth.d_flat(2, 0, 20, 29)).to("cuda:%d" % dec._device)
    for kernel, code in (("wave", 1), ("group", 2)):
        monkeypatch.setenv("CTCDEC_BEAM_KERNEL", kernel)
        dec.decode_beams(x)
        assert dec.last_beam_kernel == code
    monkeypatch.delenv("CTCDEC_BEAM_KERNEL")
    dec.decode_batch(None, [x] * 8)
    assert dec.last_beam_kernel == 2  # a handful of utterances: one workgroup each
    if torch.cuda.device_count() > 1:
        other = torch.from_numpy(synth.d_flat(2, 0, 20, 29)).to("cuda:%d" % ((dec._device + 1) % torch.cuda.device_count()))
        with pytest.raises(ValueError):
            dec.decode_beams(other)


def test_hip_permuted_and_mixed_dtype_device_batches(lm):
    """Layout / dtype conversions of device tensors run on torch's stream; the native call must only start after
    them (the synchronisation used to come BEFORE the conversions)."""
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(synth.LIBRI_LABELS, lm.path)
    xs = np.stack([synth.d_words(2, u, 300, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=6.0) for u in range(24)])
    want = dec.decode_batch(None, [x for x in xs])
    tbv = torch.from_numpy(xs).cuda().permute(1, 0, 2).contiguous()      # [T, B, V] as a model would hold it
    assert dec.decode_batch(None, tbv.permute(1, 0, 2)) == want          # non-contiguous [B, T, V] view
    mixed = [torch.from_numpy(x).cuda().to(torch.float64 if u % 2 else torch.float32) for u, x in enumerate(xs)]
    assert dec.decode_batch(None, mixed) == want


def test_hip_peaky_posteriors_single_label_runs(lm, bpe, monkeypatch):
    """Real-posterior-like logits (synth.d_peaky: most frames have one survivor, the label every beam ends in): both
    kernels consume such frames in runs (label_run). Same beams as the oracle; identical output with the runs
    switched off; fp32 device input; long utterances so that runs span look-ahead windows."""
    import torch

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.alphabet import Alphabet

    hot = lm.hotwords(6, 2)
    for labels, is_bpe, boost in ((bpe, True, 16.0), (synth.LIBRI_LABELS, False, 12.0)):
        dec = build_ctcdecoder(labels, lm.path)
        alpha = Alphabet.build_alphabet(labels)
        orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
        xs = [synth.d_peaky(9, u, 150 + 40 * u, labels, is_bpe, lm.words, lm.sentences, len(labels), boost=boost) for u in range(3)]
        # a long all-blank stretch: one run across several 64-frame windows, to the end of the utterance
        tail = np.full((200, xs[0].shape[1]), -8.0, dtype=np.float32)
        tail[:, len(labels)] = 8.0
        xs.append(np.concatenate([xs[0][:60], tail]))
        kw = {"hotwords": hot, "prune_history": True}
        got = dec.decode_beams_batch(None, [torch.from_numpy(x).cuda() for x in xs], **kw)
        for u, x in enumerate(xs):
            exp = _oracle_expected(orc, _f64(x), kw)
            check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got[u]], exp, what="peaky%d" % u, **_tol(x))
        monkeypatch.setenv("CTCDEC_NO_LABEL_RUNS", "1")
        plain = dec.decode_beams_batch(None, [torch.from_numpy(x).cuda() for x in xs], **kw)
        monkeypatch.delenv("CTCDEC_NO_LABEL_RUNS")
        for g, q in zip(got, plain):
            assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in g] == \
                   [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in q]


def test_hip_flat_model_file_round_trip(lm, tmp_path):
    """NgramModel(arpa).save_flat -> *.ctcdec -> build_ctcdecoder / save_to_dir / load_from_dir on the HIP build:
    same beams, scores and LM states as the decoder built from the ARPA text; damaged and kenlm-binary files are refused."""
    import torch

    from pyctcdecode_amd import BeamSearchDecoderCTC, build_ctcdecoder
    from pyctcdecode_amd.language_model import NgramModel, load_unigram_set_from_arpa

    flat = str(tmp_path / "model.ctcdec")
    NgramModel(lm.path).save_flat(flat)
    uni = sorted(load_unigram_set_from_arpa(lm.path))
    a = build_ctcdecoder(synth.LIBRI_LABELS, lm.path, uni)
    b = build_ctcdecoder(synth.LIBRI_LABELS, flat, uni)
    xs = [synth.d_words(2, u, 80, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=5.0) for u in range(3)]
    for x in xs:
        xd = torch.from_numpy(x).cuda()
        ra, rb = a.decode_beams(xd, beam_width=40), b.decode_beams(xd, beam_width=40)
        assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in ra] == [
            (o.text, o.text_frames, o.logit_score, o.lm_score) for o in rb]
        assert ra[0].last_lm_state.state == rb[0].last_lm_state.state
    d = tmp_path / "dec"
    d.mkdir()
    b.save_to_dir(str(d))
    c = BeamSearchDecoderCTC.load_from_dir(str(d))
    assert c.decode_batch(None, xs) == a.decode_batch(None, xs)
    raw = open(flat, "rb").read()
    bad = tmp_path / "cut.ctcdec"
    bad.write_bytes(raw[: len(raw) // 2])
    with pytest.raises(OSError):
        NgramModel(str(bad))
    with pytest.raises(NotImplementedError):  # kenlm's own binaries stay refused (no sample to pin a reader against)
        NgramModel(str(tmp_path / "model.binary"))


def test_hip_small_batches_choose_their_beam_kernel_by_the_input(lm):
    """Up to two utterances per CU the launcher looks at what the prune stage counted (api.cpp: one read-back between the two
    stages; backend_hip.hip: wave_kernel_chosen): real-posterior-like input (about one survivor a frame) runs on the wave
    kernel, the bench-like input (about six) on the workgroup kernel -- and both kernels return the same beams either way."""
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    _loaded_native()
    if os.environ.get("CTCDEC_BEAM_KERNEL"):
        pytest.skip("the kernel is forced in this run")
    labels = synth.LIBRI_LABELS
    dec = build_ctcdecoder(labels, lm.path)
    peaky = torch.from_numpy(synth.d_peaky(7, 0, 300, labels, False, lm.words, lm.sentences, 28)).cuda()
    flat = torch.from_numpy(synth.d_flat(2, 0, 120, 29)).cuda()
    out_p = dec.decode_beams(peaky, beam_width=50)
    assert dec.last_beam_kernel == 1, "peaky posteriors: the wave kernel"
    out_f = dec.decode_beams(flat, beam_width=50)
    assert dec.last_beam_kernel == 2, "flat logits (29 survivors a frame): the workgroup kernel"
    as_t = lambda beams: [(b.text, tuple(b.text_frames), b.logit_score, b.lm_score) for b in beams]  # noqa: E731
    for x, got in ((peaky, out_p), (flat, out_f)):
        for forced, code in (("wave", 1), ("group", 2)):
            os.environ["CTCDEC_BEAM_KERNEL"] = forced
            try:
                again = dec.decode_beams(x, beam_width=50)
                assert dec.last_beam_kernel == code
            finally:
                del os.environ["CTCDEC_BEAM_KERNEL"]
            assert as_t(again) == as_t(got), forced
