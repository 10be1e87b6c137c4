"""CPU (sim backend): save_to_dir / load_from_dir round trips (reference tests/test_decoder.py:807-931)."""
import os

import numpy as np
import pytest

from tests.golden_util import TOY_ARPA, load_cases
from tests.sim_util import sim_library  # noqa: F401

CASES, INPUTS = load_cases()
BY_NAME = {c["name"]: c for c in CASES}
LABELS = BY_NAME["toy_nolm_16beams"]["labels"]
X = INPUTS[BY_NAME["toy_nolm_16beams"]["input"]]


def test_round_trip_with_lm(tmp_path, sim_library):  # noqa: F811
    from pyctcdecode_amd import BeamSearchDecoderCTC, build_ctcdecoder

    dec = build_ctcdecoder(LABELS, TOY_ARPA, ["bugs", "bunny", "zzz"], alpha=0.7, beta=2.0)
    d = tmp_path / "dec"
    d.mkdir()
    dec.save_to_dir(str(d))
    assert sorted(os.listdir(d)) == ["alphabet.json", "language_model"]
    assert sorted(os.listdir(d / "language_model")) == ["attrs.json", "bugs_bunny_kenlm.arpa", "unigrams.txt"]
    assert (d / "language_model" / "unigrams.txt").read_text().split() == ["bugs", "bunny"]  # "zzz" is not in the LM
    back = BeamSearchDecoderCTC.load_from_dir(str(d))
    lm = back._language_model
    assert (lm.alpha, lm.beta, lm.unk_score_offset, lm.score_boundary) == (0.7, 2.0, -10.0, True)
    a, b = dec.decode_beams(X), back.decode_beams(X)
    assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in a] == [
        (o.text, o.text_frames, o.logit_score, o.lm_score) for o in b]


def test_round_trip_without_lm_and_errors(tmp_path, sim_library):  # noqa: F811
    from pyctcdecode_amd import BeamSearchDecoderCTC, LanguageModel, build_ctcdecoder

    dec = build_ctcdecoder(LABELS)
    d = tmp_path / "dec"
    d.mkdir()
    dec.save_to_dir(str(d))
    assert os.listdir(d) == ["alphabet.json"]
    back = BeamSearchDecoderCTC.load_from_dir(str(d))
    assert back.decode(X) == dec.decode(X)
    (d / "stray.txt").write_text("x")
    with pytest.raises(ValueError):
        BeamSearchDecoderCTC.load_from_dir(str(d))
    empty = tmp_path / "lm"
    empty.mkdir()
    with pytest.raises(ValueError):
        LanguageModel.load_from_dir(str(empty))


def test_flat_model_file_round_trip(tmp_path, sim_library):  # noqa: F811
    """ARPA -> *.ctcdec (the parsed tables) -> identical decodes, identical scorer answers; a decoder
    directory may hold the flat file in place of the ARPA file; damaged files are refused."""
    import synth
    from pyctcdecode_amd import BeamSearchDecoderCTC, build_ctcdecoder
    from pyctcdecode_amd.language_model import LanguageModel, NgramModel, load_unigram_set_from_arpa
    from tests.golden_util import LM_DIR

    lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
    flat = str(tmp_path / "model.ctcdec")
    src = NgramModel(lm.path)
    src.save_flat(flat)
    with pytest.raises(ValueError):
        src.save_flat(str(tmp_path / "model.bin"))
    back = NgramModel(flat)
    assert back.order == src.order == 4
    for w in lm.words[:50] + ["<s>", "</s>", "<unk>", "nope"]:
        assert back.index(w) == src.index(w)
    uni = sorted(load_unigram_set_from_arpa(lm.path))
    a = build_ctcdecoder(synth.LIBRI_LABELS, lm.path, uni)
    b = build_ctcdecoder(synth.LIBRI_LABELS, flat, uni)
    xs = [synth.d_words(2, u, 60, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=5.0) for u in range(3)]
    for x in xs:
        ra, rb = a.decode_beams(x, beam_width=30), b.decode_beams(x, beam_width=30)
        assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in ra] == [
            (o.text, o.text_frames, o.logit_score, o.lm_score) for o in rb]
        assert ra[0].last_lm_state.state == rb[0].last_lm_state.state
    # decoder directory with the flat file as the model file
    d = tmp_path / "dec"
    d.mkdir()
    b.save_to_dir(str(d))
    assert sorted(os.listdir(d / "language_model")) == ["attrs.json", "model.ctcdec", "unigrams.txt"]
    c = BeamSearchDecoderCTC.load_from_dir(str(d))
    assert c.decode(xs[0]) == a.decode(xs[0])
    # truncated / foreign files
    raw = open(flat, "rb").read()
    bad = tmp_path / "cut.ctcdec"
    bad.write_bytes(raw[: len(raw) // 2])
    with pytest.raises(OSError):
        NgramModel(str(bad))
    bad.write_bytes(b"mmap lm http://kheafield.com/code format version 5\n\0" + raw[60:])
    with pytest.raises(OSError):
        NgramModel(str(bad))
    with pytest.raises(NotImplementedError):
        NgramModel(str(tmp_path / "model.bin"))
    assert isinstance(c._language_model, LanguageModel)
