"""CPU: label normalisation tables of the reference (tests/test_alphabet.py:13-47 there)."""
import json

import pytest

from pyctcdecode_amd.alphabet import Alphabet, _normalize_bpe_alphabet, _normalize_regular_alphabet

KNOWN = [
    ([" ", "a", "b"], [" ", "a", "b", ""], False),
    (["<pad>", "<s>", "</s>", "<unk>", "|", "A", "B"], ["", "<s>", "</s>", "⁇", " ", "A", "B"], False),
    (["<unk>", "▁", "##a", "##b", "a", "b"], ["▁⁇▁", "▁", "a", "b", "▁a", "▁b", ""], True),
]


def test_known_mappings():
    for labels, expected, is_bpe in KNOWN:
        a = Alphabet.build_alphabet(labels)
        assert a.labels == expected and a.is_bpe == is_bpe
        b = Alphabet.loads(a.dumps())
        assert (b.labels, b.is_bpe) == (a.labels, a.is_bpe)


def test_rules():
    assert _normalize_regular_alphabet([" ", "a", "b", ""]) == [" ", "a", "b", ""]
    assert _normalize_regular_alphabet(["_", "a", " "]) == ["", "a", " "]
    bpe = ["▁⁇▁", "▁", "a", "b", "▁a", "▁b"]
    assert _normalize_bpe_alphabet(bpe) == bpe + [""]
    assert _normalize_bpe_alphabet(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "##a", "##b", "a", "b"]) == [
        "", "▁⁇▁", "[CLS]", "[SEP]", "[MASK]", "a", "b", "▁a", "▁b"]


def test_errors():
    with pytest.raises(ValueError):
        Alphabet.build_alphabet(["a", "a", "b"])
    with pytest.raises(ValueError):
        Alphabet.build_alphabet(["▁a", " "])
    for bad in ({"labels": ["a"]}, {"labels": ["a"], "is_bpe": True, "extra": 1}):
        with pytest.raises(ValueError):
            Alphabet.loads(json.dumps(bad))
