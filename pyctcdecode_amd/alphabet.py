"""Label normalisation for the decoder (host side, build time only).

Mirrors the behaviour of the reference's pyctcdecode/alphabet.py (:22-31 BPE detection, :34-73
char vocabularies, :76-110 BPE vocabularies, :113-120 verification, :139-162 build/dumps/loads,
:165-170 coverage warning) and its tables in tests/test_alphabet.py:13-47.  The normalised label
list defines the per-token classes the device kernels consume: blank (""), word separator (" "
for char vocabularies, a leading U+2581 for BPE), trailing U+2581 (forces the next word break).
"""
from __future__ import annotations

import json
import logging
import re
from typing import Collection, List

BPE_TOKEN = "▁"
UNK_TOKEN = "⁇"
UNK_BPE_TOKEN = "▁⁇▁"

_SPECIAL = re.compile(r"^[<\[].+[>\]]$")
_BLANK = re.compile(r"^[<\[]pad[>\]]$", flags=re.IGNORECASE)
_UNK = re.compile(r"^[<\[]unk[>\]]$", flags=re.IGNORECASE)

logger = logging.getLogger(__name__)


def _check_if_bpe(labels: List[str]) -> bool:
    is_bpe = any(s.startswith("##") for s in labels) or any(s.startswith(BPE_TOKEN) for s in labels)
    logger.info("Alphabet determined to be of %s style.", "BPE" if is_bpe else "regular")
    return is_bpe


def _substitute(labels: List[str], pattern, replacement: str, what: str) -> None:
    for n, label in enumerate(labels):
        if pattern.match(label):
            logger.info("Found %s in vocabulary, interpreted as %s, substituting with %r.", label, what, replacement)
            labels[n] = replacement


def _normalize_regular_alphabet(labels: List[str]) -> List[str]:
    out = list(labels)
    if "|" in out and " " not in out:
        logger.info("Found '|' in vocabulary but not ' ', doing substitution.")
        out[out.index("|")] = " "
    _substitute(out, _BLANK, "", "a CTC blank token")
    if "_" in out and "" not in out:
        logger.info("Found '_' in vocabulary but not '', doing substitution.")
        out[out.index("_")] = ""
    if "" not in out:
        logger.info("CTC blank char '' not found, appending to end.")
        out.append("")
    _substitute(out, _UNK, UNK_TOKEN, "unknown token")
    if any(len(c) > 1 for c in out):
        logger.warning(
            "Found entries of length > 1 in alphabet. This is unusual unless style is BPE, but the "
            "alphabet was not recognized as BPE type. Is this correct?"
        )
    if " " not in out:
        logger.warning("Space token ' ' missing from vocabulary.")
    return out


def _convert_bpe_token_style(token: str) -> str:
    """'##x' continuation style -> U+2581 word-start style."""
    if token.startswith("##"):
        return token[2:]
    if _SPECIAL.match(token) or token in ("", BPE_TOKEN, UNK_BPE_TOKEN, "<unk>"):
        return token
    return BPE_TOKEN + token


def _normalize_bpe_alphabet(labels: List[str]) -> List[str]:
    out = list(labels)
    if any(s.startswith("##") for s in labels):
        out = [_convert_bpe_token_style(c) for c in out]
    _substitute(out, _BLANK, "", "a CTC blank token")
    if "" not in out:
        logger.info("CTC blank char '' not found, appending to end.")
        out.append("")
    _substitute(out, _UNK, UNK_BPE_TOKEN, "unknown token")
    if UNK_BPE_TOKEN not in out:
        logger.warning("UNK token %s not found, is this a mistake?", UNK_BPE_TOKEN)
    return out


def _verify_alphabet(labels: List[str], is_bpe: bool) -> None:
    if len(labels) != len(set(labels)):
        raise ValueError("Alphabet contains duplicate entries, this is not allowed.")
    if is_bpe and any(" " in s for s in labels):
        raise ValueError("Space token ' ' found in vocabulary even though it looks like BPE.")


class Alphabet:
    def __init__(self, labels: List[str], is_bpe: bool) -> None:
        self._labels = labels
        self._is_bpe = is_bpe

    @property
    def is_bpe(self) -> bool:
        return self._is_bpe

    @property
    def labels(self) -> List[str]:
        return self._labels[:]

    @classmethod
    def build_alphabet(cls, labels: List[str]) -> "Alphabet":
        is_bpe = _check_if_bpe(labels)
        _verify_alphabet(labels, is_bpe)
        norm = _normalize_bpe_alphabet(labels) if is_bpe else _normalize_regular_alphabet(labels)
        return cls(norm, is_bpe)

    def dumps(self) -> str:
        return json.dumps({"labels": self.labels, "is_bpe": self.is_bpe})

    @classmethod
    def loads(cls, s: str) -> "Alphabet":
        d = json.loads(s)
        expected = {"is_bpe", "labels"}
        if set(d.keys()) != expected:
            raise ValueError(f"unexpected keys found. Expected {expected}, found {set(d.keys())}")
        return cls(d["labels"], d["is_bpe"])


def verify_alphabet_coverage(alphabet: Alphabet, unigrams: Collection[str]) -> None:
    label_chars = set(alphabet.labels)
    unigram_chars = set("".join(unigrams))
    if len(unigram_chars - label_chars) / len(unigram_chars) > 0.2:
        logger.warning("Unigrams and labels don't seem to agree.")
