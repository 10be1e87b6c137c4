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
