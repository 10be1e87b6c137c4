"""kenlm PROBING binaries (csrc/kenlm_binary.cpp; SURVEY 8(f) rank 2: `kenlm_model_path="x.bin"`, decoder.py:1074,
language_model.py:424).

FORMAT UNPINNED AGAINST REAL KENLM: neither kenlm nor a file it wrote exists in this container. What is pinned here is the
reader against this library's own writer of the layout kenlm's sources describe (ARPA -> ctcdec_arpa_to_kenlm_binary -> reader):
the loaded model must be the ARPA-built one -- same word indices, same score and same out-state for EVERY n-gram of the file and
for back-off queries -- and a decoder built on the .bin must decode like the one built on the .arpa. Plus what a reader of
somebody else's format owes its users: every refusal is by name, and a file whose layout is not what the reader assumes fails
loudly (vocabulary strings are checked against the vocabulary hash table) instead of scoring wrongly."""
import os
import struct

import numpy as np
import pytest

import synth
from tests.golden_util import GOLD, LM_DIR, TOY_ARPA
from tests.sim_util import sim_library  # noqa: F401

KNOWN = os.path.join(GOLD, "ngram_known.arpa")
KNOWN_NOUNK = os.path.join(GOLD, "ngram_known_nounk.arpa")


def _arpa_ngrams(path):
    """[(words tuple, prob, backoff)] of every order, as listed"""
    out, section = [], 0
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line.startswith("\\") and line.endswith("-grams:"):
                section = int(line[1:line.index("-")])
                continue
            if not line or line.startswith("\\") or section == 0:
                continue
            parts = line.split("\t")
            out.append((tuple(parts[1].split(" ")), float(parts[0]), float(parts[2]) if len(parts) > 2 and parts[2] else 0.0))
    return out


def _score_all(model, grams):
    """(log10 p, out state) of every listed n-gram's last word after its context, and of the same word after an unseen context"""
    from pyctcdecode_amd.language_model import NgramState

    res = []
    for words, _p, _b in grams:
        st = model.start_state(False)
        for w in words[:-1]:
            nxt = NgramState()
            model.BaseScore(st, w, nxt)
            st = nxt
        out = NgramState()
        res.append((words, model.BaseScore(st, words[-1], out), out.length, out.words, out.backoff))
        # a back-off query: the same word behind a context that starts with an out-of-vocabulary word
        st2 = model.start_state(True)
        nxt = NgramState()
        model.BaseScore(st2, "never-seen-word", nxt)
        out2 = NgramState()
        res.append((("<oov>",) + words[-1:], model.BaseScore(nxt, words[-1], out2), out2.length, out2.words, out2.backoff))
    return res


def _convert(tmp_path, arpa, multiplier=1.5, name="m.bin"):
    from pyctcdecode_amd.language_model import NgramModel

    out = str(tmp_path / name)
    NgramModel.arpa_to_kenlm_binary(arpa, out, multiplier)
    return out


def _synth_arpa():
    return synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2).path


@pytest.mark.parametrize("which", ["toy", "known", "known_nounk", "synth"])
@pytest.mark.parametrize("multiplier", [1.5, 2.25])
def test_binary_model_equals_the_arpa_model(which, multiplier, tmp_path, sim_library):  # noqa: F811
    from pyctcdecode_amd.language_model import NgramModel

    arpa = {"toy": TOY_ARPA, "known": KNOWN, "known_nounk": KNOWN_NOUNK, "synth": _synth_arpa}[which]
    arpa = arpa() if callable(arpa) else arpa
    path = _convert(tmp_path, arpa, multiplier)
    a, b = NgramModel(arpa), NgramModel(path)
    assert a.order == b.order
    grams = _arpa_ngrams(arpa)
    if which == "synth":
        grams = grams[::7]
    for words, _p, _b in grams:
        for w in words:
            assert a.index(w) == b.index(w) and (a.index(w) == 0 or b.word(b.index(w)) == w)
    assert b.index("never-seen-word") == 0 and ("<unk>" in b) is False
    sa, sb = _score_all(a, grams), _score_all(b, grams)
    assert sa == sb  # bit for bit: the same float32 probabilities and back-offs, the same states
    assert a.start_state(True) == b.start_state(True) and a.start_state(False) == b.start_state(False)


def test_decoder_on_a_binary_decodes_like_the_arpa_one(tmp_path, sim_library, both_beam_kernels):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.language_model import load_unigram_set_from_arpa

    lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
    path = _convert(tmp_path, lm.path, name="model.binary")
    uni = load_unigram_set_from_arpa(lm.path)
    da = build_ctcdecoder(synth.LIBRI_LABELS, lm.path, uni)
    db = build_ctcdecoder(synth.LIBRI_LABELS, path, uni)
    for u in range(3):
        x = synth.d_words(2, u, 60, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=3.0).astype(np.float64)
        ba, bb = da.decode_beams(x, beam_width=32, beam_prune_logp=-40.0), db.decode_beams(x, beam_width=32, beam_prune_logp=-40.0)
        assert len(ba) > 3
        assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in ba] == [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in bb]
    # without a unigram list a binary model decodes with the reference's warning (decoder.py:1081-1084), not an error
    dn = build_ctcdecoder(synth.LIBRI_LABELS, path)
    assert isinstance(dn.decode(x), str)


@pytest.mark.gpu
def test_hip_decoder_on_a_binary_decodes_like_the_arpa_one(tmp_path, both_beam_kernels):
    import torch

    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.language_model import load_unigram_set_from_arpa

    lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
    path = _convert(tmp_path, lm.path, name="model.bin")
    uni = load_unigram_set_from_arpa(lm.path)
    da = build_ctcdecoder(synth.LIBRI_LABELS, lm.path, uni)
    db = build_ctcdecoder(synth.LIBRI_LABELS, path, uni)
    xs = [torch.from_numpy(synth.d_words(2, u, 80, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=3.0)).cuda() for u in range(4)]
    ba, bb = da.decode_beams_batch(None, xs, beam_width=32, beam_prune_logp=-40.0), db.decode_beams_batch(None, xs, beam_width=32, beam_prune_logp=-40.0)
    for p, q in zip(ba, bb):
        assert len(p) > 3
        assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in p] == [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in q]


def _expect_refusal(path, needle):
    from pyctcdecode_amd import _binding as B
    from pyctcdecode_amd.language_model import NgramModel

    with pytest.raises((B.NativeError, OSError, ValueError)) as e:
        NgramModel(path)
    assert needle in str(e.value), str(e.value)


def test_refusals_are_by_name_and_misread_layouts_fail_loudly(tmp_path, sim_library):  # noqa: F811
    good = _convert(tmp_path, KNOWN)
    blob = bytearray(open(good, "rb").read())

    def variant(name, edit):
        b = bytearray(blob)
        edit(b)
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(b))
        return p

    # model types (FixedWidthParameters.model_type at byte 96): everything but probing is named and refused
    for code, name in ((1, "rest-cost probing"), (2, "trie"), (3, "quantised trie"), (4, "array-compressed trie"), (5, "quantised array-compressed trie")):
        _expect_refusal(variant("t%d.bin" % code, lambda b, c=code: b.__setitem__(slice(96, 100), struct.pack("<i", c))), "'%s'" % name)
    _expect_refusal(variant("v4.bin", lambda b: b.__setitem__(49, ord("4"))), "format version 4")
    _expect_refusal(variant("novocab.bin", lambda b: b.__setitem__(100, 0)), "without its vocabulary strings")
    _expect_refusal(variant("sanity.bin", lambda b: b.__setitem__(slice(60, 64), struct.pack("<f", 2.0))), "sanity block")
    _expect_refusal(variant("short.bin", lambda b: b.__delitem__(slice(len(b) - 40, len(b)))), "truncated")
    # a vocabulary string that is not what the hash table says (a stand-in for any layout the reader misplaces by a byte)
    last = bytes(blob).rstrip(b"\0").rfind(b"\0") + 1
    _expect_refusal(variant("word.bin", lambda b: b.__setitem__(last, b[last] ^ 1)), "not in the vocabulary hash table")
    # an ARPA file is still an ARPA file, and any other unknown file is refused before the library sees it
    other = str(tmp_path / "x.bin")
    open(other, "wb").write(b"not a language model at all, but long enough to be sniffed............")
    from pyctcdecode_amd.language_model import NgramModel

    with pytest.raises(NotImplementedError):
        NgramModel(other)


def test_header_layout_is_the_documented_one(tmp_path, sim_library):  # noqa: F811
    """The bytes the writer emits, field by field (the documentation in csrc/kenlm_binary.cpp, as a test)."""
    path = _convert(tmp_path, KNOWN)
    b = open(path, "rb").read()
    assert b[:52] == b"mmap lm http://kheafield.com/code format version 5\n\0" and b[52:56] == b"\0\0\0\0"
    assert struct.unpack_from("<fffII", b, 56) == (0.0, 1.0, -0.5, 1, 0xFFFFFFFF) and struct.unpack_from("<Q", b, 80) == (1,)
    order = b[88]
    mult, mtype, has_vocab, sver = struct.unpack_from("<f", b, 92)[0], struct.unpack_from("<i", b, 96)[0], b[100], struct.unpack_from("<I", b, 104)[0]
    assert (order, mult, mtype, has_vocab, sver) == (4, 1.5, 0, 1, 0)
    counts = struct.unpack_from("<%dQ" % order, b, 108)
    listed = _arpa_ngrams(KNOWN)
    assert list(counts) == [sum(1 for g in listed if len(g[0]) == n) for n in range(1, order + 1)]
    pos = (108 + 8 * order + 7) & ~7
    bound = struct.unpack_from("<II", b, pos)[1]
    assert bound == counts[0]  # (<unk> is listed in this file)
    vb = max(counts[0] + 1, int(np.float32(1.5) * np.float32(counts[0])))
    pos += 8 + vb * 12 + (counts[0] + 1) * 8
    for n in range(2, order + 1):
        pos += max(counts[n - 1] + 1, int(np.float32(1.5) * np.float32(counts[n - 1]))) * (12 if n == order else 16)
    words = b[pos:].split(b"\0")
    assert words[0] == b"<unk>" and words[-1] == b"" and len(words) - 1 == bound
