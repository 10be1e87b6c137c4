"""Label normalisation for the decoder (host side, build time only).

Behaviour follows the reference's pyctcdecode/alphabet.py (BPE detection :22-31, char
vocabularies :34-73, BPE vocabularies :76-110, verification :113-120, build/dumps/loads :139-162,
coverage warning :165-170) and its tables in tests/test_alphabet.py:13-47.  The normalised list
defines the per-label classes the device kernels consume: blank "", word separator (" " for char
vocabularies, a leading U+2581 for BPE), trailing U+2581 (forces the next word break).

Implementation: a vocabulary is pushed through a short list of rewrite rules; each rule is a small
function from label list to label list.
"""
from __future__ import annotations

import json
import logging
import re
from typing import Callable, Collection, List, Sequence

BPE_TOKEN = "▁"
UNK_TOKEN = "⁇"
UNK_BPE_TOKEN = BPE_TOKEN + UNK_TOKEN + BPE_TOKEN

logger = logging.getLogger(__name__)

_BRACKETED = re.compile(r"^[<\[].+[>\]]$")          # <s>, [CLS], ...
_PAD = re.compile(r"^[<\[]pad[>\]]$", re.IGNORECASE)  # CTC blank spelled as a pad token
_UNK = re.compile(r"^[<\[]unk[>\]]$", re.IGNORECASE)

# names under which callers of the reference import the same patterns (alphabet.py:15-17)
SPECIAL_TOKEN_PTN, BLANK_TOKEN_PTN, UNK_TOKEN_PTN = _BRACKETED, _PAD, _UNK

Rule = Callable[[List[str]], List[str]]


def _check_if_bpe(labels: Sequence[str]) -> bool:
    """A vocabulary is BPE style when some piece carries a '##' or U+2581 prefix."""
    found = any(s.startswith(("##", BPE_TOKEN)) for s in labels)
    logger.info("alphabet style: %s", "BPE" if found else "regular")
    return found


def _rename_matching(pattern, new: str) -> Rule:
    def rule(labels: List[str]) -> List[str]:
        return [new if pattern.match(lab) else lab for lab in labels]

    return rule


def _rename_if_target_absent(old: str, new: str) -> Rule:
    """old -> new for the first `old`, but only when `new` is not a label already."""

    def rule(labels: List[str]) -> List[str]:
        if old in labels and new not in labels:
            labels = list(labels)
            labels[labels.index(old)] = new
        return labels

    return rule


def _ensure_blank(labels: List[str]) -> List[str]:
    return labels if "" in labels else labels + [""]


def _wordpiece_to_sentencepiece(labels: List[str]) -> List[str]:
    """'##x' continuation style -> U+2581 word-start style, applied when any '##' piece exists."""
    if not any(s.startswith("##") for s in labels):
        return labels
    return [_convert_bpe_token_style(s) for s in labels]


def _convert_bpe_token_style(token: str) -> str:
    if token.startswith("##"):
        return token[2:]
    keep_as_is = _BRACKETED.match(token) or token in ("", BPE_TOKEN, UNK_BPE_TOKEN, "<unk>")
    return token if keep_as_is else BPE_TOKEN + token


_REGULAR_RULES: Sequence[Rule] = (
    _rename_if_target_absent("|", " "),
    _rename_matching(_PAD, ""),
    _rename_if_target_absent("_", ""),
    _ensure_blank,
    _rename_matching(_UNK, UNK_TOKEN),
)
_BPE_RULES: Sequence[Rule] = (
    _wordpiece_to_sentencepiece,
    _rename_matching(_PAD, ""),
    _ensure_blank,
    _rename_matching(_UNK, UNK_BPE_TOKEN),
)


def _apply(rules: Sequence[Rule], labels: Sequence[str]) -> List[str]:
    out = list(labels)
    for rule in rules:
        out = rule(out)
    return out


def _normalize_regular_alphabet(labels: List[str]) -> List[str]:
    out = _apply(_REGULAR_RULES, labels)
    if any(len(c) > 1 for c in out):
        logger.warning("labels longer than one character in a vocabulary that does not look like BPE")
    if " " not in out:
        logger.warning("no word separator ' ' in the vocabulary")
    return out


def _normalize_bpe_alphabet(labels: List[str]) -> List[str]:
    out = _apply(_BPE_RULES, labels)
    if UNK_BPE_TOKEN not in out:
        logger.warning("no unknown-token piece %s in the BPE vocabulary", UNK_BPE_TOKEN)
    return out


def _verify_alphabet(labels: List[str], is_bpe: bool) -> None:
    if len(set(labels)) != len(labels):
        raise ValueError("Alphabet contains duplicate entries, this is not allowed.")
    if is_bpe and any(" " in s for s in labels):
        raise ValueError("Space token ' ' found in vocabulary even though it looks like BPE.")


class Alphabet:
    def __init__(self, labels: List[str], is_bpe: bool) -> None:
        self._labels, self._is_bpe = labels, is_bpe

    @property
    def is_bpe(self) -> bool:
        return self._is_bpe

    @property
    def labels(self) -> List[str]:
        return list(self._labels)

    @classmethod
    def build_alphabet(cls, labels: List[str]) -> "Alphabet":
        bpe = _check_if_bpe(labels)
        _verify_alphabet(labels, bpe)
        return cls((_normalize_bpe_alphabet if bpe else _normalize_regular_alphabet)(labels), bpe)

    def dumps(self) -> str:
        return json.dumps({"labels": self.labels, "is_bpe": self.is_bpe})

    @classmethod
    def loads(cls, s: str) -> "Alphabet":
        d = json.loads(s)
        if set(d) != {"is_bpe", "labels"}:
            raise ValueError("unexpected keys found. Expected {'is_bpe', 'labels'}, found %s" % sorted(d))
        return cls(d["labels"], d["is_bpe"])


def verify_alphabet_coverage(alphabet: Alphabet, unigrams: Collection[str]) -> None:
    """Warn when more than a fifth of the unigrams' characters are not labels."""
    chars = set("".join(unigrams))
    if len(chars - set(alphabet.labels)) / len(chars) > 0.2:
        logger.warning("Unigrams and labels don't seem to agree.")
