"""Scorer objects of the drop-in API (reference: pyctcdecode/language_model.py).

``NgramModel`` plays the role of ``kenlm.Model`` (an external C++ dependency of the reference):
it owns the flat hashed n-gram trie inside libctcdec.  ``LanguageModel`` mirrors the reference's
wrapper (language_model.py:230-360) for the public ``score`` / ``score_partial_token`` /
``get_start_state`` methods; the decode path itself never calls these from Python -- the same
arithmetic runs on the device (csrc/beam_core.h) with alpha/beta/unk/boundary as kernel arguments.
"""
from __future__ import annotations

import abc
import ctypes as C
import logging
import os
import re
from typing import Any, Collection, Dict, Iterable, List, Optional, Sequence, Set, Tuple

from . import _binding as B
from .constants import (
    AVG_TOKEN_LEN,
    DEFAULT_ALPHA,
    DEFAULT_BETA,
    DEFAULT_HOTWORD_WEIGHT,
    DEFAULT_SCORE_LM_BOUNDARY,
    DEFAULT_UNK_LOGP_OFFSET,
    LOG_BASE_CHANGE_FACTOR,
)

logger = logging.getLogger(__name__)


class AbstractLMState(abc.ABC):
    """language_model.py:37-42."""

    def get_mp_safe_state(self) -> Optional["AbstractLMState"]:
        return None


class NgramState:
    """Context words (newest first, as LM vocabulary indices) + back-off weights. kenlm.State's role."""

    __slots__ = ("length", "words", "backoff")

    def __init__(self, length: int = 0, words: Sequence[int] = (), backoff: Sequence[float] = ()):
        self.length = length
        self.words = tuple(words)
        self.backoff = tuple(backoff)

    @classmethod
    def from_c(cls, st: B.LmState) -> "NgramState":
        n = max(0, st.length)
        return cls(st.length, [st.words[k] for k in range(n)], [st.backoff[k] for k in range(n)])

    def to_c(self) -> B.LmState:
        st = B.LmState()
        st.length = self.length
        for k in range(max(0, self.length)):
            st.words[k] = self.words[k]
            st.backoff[k] = self.backoff[k]
        return st

    def __eq__(self, other):
        return isinstance(other, NgramState) and (self.length, self.words, self.backoff) == (
            other.length, other.words, other.backoff)

    def __repr__(self):
        return "NgramState(words=%r, backoff=%r)" % (self.words, self.backoff)


class KenlmState(AbstractLMState):
    """language_model.py:45-53."""

    def __init__(self, state: NgramState) -> None:
        self._state = state

    @property
    def state(self) -> NgramState:
        return self._state


class MultiLanguageModelState(AbstractLMState):
    """language_model.py:56-64: one state per contained model."""

    def __init__(self, states: List[AbstractLMState]) -> None:
        self._states = states

    @property
    def states(self) -> List[AbstractLMState]:
        return self._states


class NgramModel:
    """The n-gram model inside libctcdec: what ``kenlm.Model(path)`` is to the reference
    (decoder.py:1074).  Reads ARPA text, kenlm PROBING binaries (``build_binary probing``; the format is restated from
    kenlm's published sources and pinned only against this library's own writer: csrc/kenlm_binary.cpp) and its own flat model
    files (``*.ctcdec``, written by :meth:`save_flat`: the ARPA file parsed once, loading is a few reads)."""

    @staticmethod
    def arpa_to_kenlm_binary(arpa_path: str, out_path: str, probing_multiplier: float = 1.5) -> None:
        """ARPA -> kenlm probing binary (the layout ``build_binary probing`` writes, as far as kenlm's sources say)."""
        lib = B.get_library()
        lib.check(lib.dll.ctcdec_arpa_to_kenlm_binary(arpa_path.encode("utf-8"), out_path.encode("utf-8"), float(probing_multiplier)))

    FLAT_SUFFIX = ".ctcdec"

    def __init__(self, path: str, _clone_of: Optional["NgramModel"] = None):
        lib = B.get_library()
        self._lib = lib
        self.path = path.encode("utf-8")
        self._unigrams_owned = False  # a LanguageModel has configured this model's unigram set
        if _clone_of is not None:
            blob, off = B.pack_strings([""])
            handle = C.c_void_p()
            lib.check(lib.dll.ctcdec_create(blob, B.off_ptr(off), 1, 0, _default_device(), C.byref(handle)))
            self._handle = handle
            lib.check(lib.dll.ctcdec_lm_clone(handle, _clone_of._handle))
            self.order = _clone_of.order
            return
        flat = path.endswith(self.FLAT_SUFFIX)
        # a kenlm binary (decoder.py:1074 hands any path to kenlm.Model; language_model.py:424 accepts .bin / .binary): told by
        # its magic bytes, whatever it is called. Only the PROBING format is read (the library refuses the others by name).
        kenlm_bin = not flat and os.path.isfile(path) and bool(lib.dll.ctcdec_is_kenlm_binary(self.path))
        if not flat and not kenlm_bin and not path.endswith(".arpa"):
            raise NotImplementedError(
                "language model files are read as ARPA text (.arpa), as kenlm probing binaries (told by their magic bytes) or as "
                "this package's flat files (%s); got %r" % (self.FLAT_SUFFIX, path)
            )
        blob, off = B.pack_strings([""])
        handle = C.c_void_p()
        lib.check(lib.dll.ctcdec_create(blob, B.off_ptr(off), 1, 0, _default_device(), C.byref(handle)))
        self._handle = handle
        order = C.c_int32()
        try:
            load = lib.dll.ctcdec_lm_load_flat if flat else (lib.dll.ctcdec_lm_load_kenlm if kenlm_bin else lib.dll.ctcdec_lm_load_arpa)
            lib.check(load(handle, self.path, C.byref(order)))
        except Exception:
            lib.dll.ctcdec_destroy(handle)
            self._handle = None
            raise
        self.order = int(order.value)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                self._lib.dll.ctcdec_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def save_flat(self, path: str) -> None:
        """Write the parsed model as one flat file; ``NgramModel(path)`` / ``build_ctcdecoder(labels, path,
        unigrams)`` load it without parsing (the name must end in ``.ctcdec``)."""
        if not path.endswith(self.FLAT_SUFFIX):
            raise ValueError("flat model files are named *%s" % self.FLAT_SUFFIX)
        self._lib.check(self._lib.dll.ctcdec_lm_save_flat(self._handle, path.encode("utf-8")))

    def index(self, word: str) -> int:
        w = word.encode("utf-8")
        out = C.c_uint32()
        self._lib.check(self._lib.dll.ctcdec_lm_word_index(self._handle, w, len(w), C.byref(out)))
        return int(out.value)

    def word(self, index: int) -> str:
        p = C.c_void_p()
        n = C.c_int64()
        self._lib.check(self._lib.dll.ctcdec_lm_word_string(self._handle, index, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value).decode("utf-8")

    def __contains__(self, word: str) -> bool:
        return self.index(word) != 0

    def start_state(self, begin_sentence: bool) -> NgramState:
        st = B.LmState()
        self._lib.check(self._lib.dll.ctcdec_lm_start_state(self._handle, int(begin_sentence), C.byref(st)))
        return NgramState.from_c(st)

    def BeginSentenceWrite(self, state: NgramState) -> None:  # noqa: N802 (kenlm name)
        s = self.start_state(True)
        state.length, state.words, state.backoff = s.length, s.words, s.backoff

    def NullContextWrite(self, state: NgramState) -> None:  # noqa: N802
        s = self.start_state(False)
        state.length, state.words, state.backoff = s.length, s.words, s.backoff

    def BaseScore(self, in_state: NgramState, word: str, out_state: NgramState) -> float:  # noqa: N802
        cin = in_state.to_c()
        cout = B.LmState()
        p = C.c_float()
        self._lib.check(
            self._lib.dll.ctcdec_lm_base_score(self._handle, C.byref(cin), self.index(word), C.byref(cout), C.byref(p))
        )
        s = NgramState.from_c(cout)
        out_state.length, out_state.words, out_state.backoff = s.length, s.words, s.backoff
        return float(p.value)

    def private_copy(self) -> "NgramModel":
        """The same n-gram tables under a handle of their own (its unigram set can differ)."""
        return NgramModel(self.path.decode("utf-8"), _clone_of=self)

    def set_unigrams(self, unigrams: Optional[Collection[str]]) -> int:
        kept = C.c_int64()
        if unigrams is None:
            self._lib.check(self._lib.dll.ctcdec_lm_set_unigrams(self._handle, 0, None, None, 0, C.byref(kept)))
        else:
            blob, off = B.pack_strings(list(unigrams))
            self._lib.check(
                self._lib.dll.ctcdec_lm_set_unigrams(self._handle, 1, blob, B.off_ptr(off), len(off) - 1, C.byref(kept))
            )
        return int(kept.value)

    def prefix_flags(self, text: str) -> int:
        t = text.encode("utf-8")
        out = C.c_uint32()
        self._lib.check(self._lib.dll.ctcdec_lm_prefix_flags(self._handle, t, len(t), C.byref(out)))
        return int(out.value)


def _default_device() -> int:
    import os

    return int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("CTCDEC_DEVICE") is None else int(
        os.environ["CTCDEC_DEVICE"])


def load_unigram_set_from_arpa(arpa_path: str) -> Set[str]:
    """The words of an ARPA file's \\1-grams: section (language_model.py:67-84): the middle field of every line there that
    has exactly three tab-separated fields (probability, word, back-off); reading stops at \\2-grams:."""
    words: Set[str] = set()
    section = 0  # 0: before the unigrams, 1: inside
    with open(arpa_path) as arpa:
        for raw in arpa:
            entry = raw.strip()
            if entry == "\\2-grams:":
                break
            if entry == "\\1-grams:":
                section = 1
                continue
            if section == 1 and entry:
                fields = entry.split("\t")
                if len(fields) == 3:
                    words.add(fields[1])
    if not words:
        raise ValueError("No unigrams found in arpa file. Something is wrong with the file.")
    return words


class HotwordScorer:
    """language_model.py:115-189 as a word set + prefix table.  The decode path ships the same
    unigram list to the device (ctcdec_set_hotwords); this class serves the public methods."""

    def __init__(self, unigrams: Sequence[str], weight: float = DEFAULT_HOTWORD_WEIGHT) -> None:
        self._unigrams = list(unigrams)
        self._words = set(self._unigrams)
        self._min_len: Dict[str, int] = {}
        for w in self._unigrams:
            for k in range(len(w) + 1):
                p = w[:k]
                if p not in self._min_len or len(w) < self._min_len[p]:
                    self._min_len[p] = len(w)
        self._weight = weight

    @property
    def unigrams(self) -> List[str]:
        return list(self._unigrams)

    @property
    def weight(self) -> float:
        return self._weight

    def __contains__(self, item: str) -> bool:
        return item in self._min_len

    def score(self, text: str) -> float:
        return self._weight * sum(1 for w in re.split(r"\s+", text) if w in self._words)

    def score_partial_token(self, token: str) -> float:
        if token in self._min_len:
            return self._weight * len(token) / self._min_len[token]
        return 0.0

    @classmethod
    def build_scorer(
        cls, hotwords: Optional[Iterable[str]] = None, weight: float = DEFAULT_HOTWORD_WEIGHT
    ) -> "HotwordScorer":
        hotwords = hotwords or []
        hotwords = [s.strip() for s in hotwords if len(s.strip()) > 0]
        unigrams: List[str] = []
        for ngram in hotwords:
            unigrams.extend(ngram.split())
        return cls(unigrams, weight)


class AbstractLanguageModel(abc.ABC):
    """language_model.py:192-227."""

    @property
    @abc.abstractmethod
    def order(self) -> int:
        raise NotImplementedError()

    @abc.abstractmethod
    def get_start_state(self) -> AbstractLMState:
        raise NotImplementedError()

    @abc.abstractmethod
    def score_partial_token(self, partial_token: str) -> float:
        raise NotImplementedError()

    @abc.abstractmethod
    def score(
        self, prev_state: AbstractLMState, word: str, is_last_word: bool = False
    ) -> Tuple[float, AbstractLMState]:
        raise NotImplementedError()

    def save_to_dir(self, filepath: str) -> None:
        raise NotImplementedError()

    @classmethod
    def load_from_dir(cls, filepath: str) -> "AbstractLanguageModel":
        raise NotImplementedError()

    def reset_params(self, **params: Dict[str, Any]) -> None:
        """Reset some of the parameters in place."""


class LanguageModel(AbstractLanguageModel):
    """language_model.py:230-360 over an :class:`NgramModel`."""

    def __init__(
        self,
        kenlm_model: Any,
        unigrams: Optional[Collection[str]] = None,
        alpha: float = DEFAULT_ALPHA,
        beta: float = DEFAULT_BETA,
        unk_score_offset: float = DEFAULT_UNK_LOGP_OFFSET,
        score_boundary: bool = DEFAULT_SCORE_LM_BOUNDARY,
    ) -> None:
        if not isinstance(kenlm_model, NgramModel):
            # a real kenlm.Model (or anything with .path): rebuild our own trie from its file
            path = getattr(kenlm_model, "path", None)
            if path is None:
                raise TypeError("kenlm_model must be an NgramModel or expose a .path")
            kenlm_model = NgramModel(path.decode("utf-8") if isinstance(path, bytes) else path)
        # The reference treats kenlm.Model as immutable and builds many LanguageModels with different unigram sets
        # on one (tests/test_decoder.py:188-280). Here the unigram set is part of the model's tables: the first
        # LanguageModel configures the model it was given, every further one works on a private copy, so that
        # earlier LanguageModels (and the decoders sharing their tables) keep their own OOV / partial-word scoring.
        if kenlm_model._unigrams_owned:
            kenlm_model = kenlm_model.private_copy()
        kenlm_model._unigrams_owned = True
        self._kenlm_model = kenlm_model
        if unigrams is None:
            logger.warning("No known unigrams provided, decoding results might be a lot worse.")
            self._has_trie = False
            self._given_unigrams = set()
            self._n_unigrams = kenlm_model.set_unigrams(None)
        else:
            if len(unigrams) < 1000:
                logger.warning(
                    "Only %s unigrams passed as vocabulary. Is this small or artificial data?", len(unigrams)
                )
            self._has_trie = True
            self._given_unigrams = set(unigrams)
            self._n_unigrams = kenlm_model.set_unigrams(self._given_unigrams)
            retained = 1.0 if len(unigrams) == 0 else self._n_unigrams / len(unigrams)
            if retained < 0.1:
                logger.warning(
                    "Only %s%% of unigrams in vocabulary found in kenlm model-- this might mean that your "
                    "vocabulary and language model are incompatible. Is this intentional?",
                    round(retained * 100, 1),
                )
        self.alpha = alpha
        self.beta = beta
        self.unk_score_offset = unk_score_offset
        self.score_boundary = score_boundary

    # -- serialisation (language_model.py:362-452): a directory with exactly three files --------------
    JSON_ATTRS = ("alpha", "beta", "unk_score_offset", "score_boundary")
    _ATTRS_SERIALIZED_FILENAME = "attrs.json"
    _UNIGRAMS_SERIALIZED_FILENAME = "unigrams.txt"

    @property
    def _unigram_set(self) -> Set[str]:
        """Unigrams that survived the filter to the LM vocabulary (language_model.py:95)."""
        return {w for w in self._given_unigrams if w in self._kenlm_model}

    @property
    def serializable_attrs(self) -> Dict[str, Any]:
        attrs = {}
        for name in LanguageModel.JSON_ATTRS:
            val = getattr(self, name)
            if val is None:
                raise ValueError(f"attribute {name} not found. Cannot serialize")
            attrs[name] = val
        return attrs

    def save_to_dir(self, filepath: str, unigram_encoding: Optional[str] = None) -> None:
        import json
        import os
        import shutil

        with open(os.path.join(filepath, self._ATTRS_SERIALIZED_FILENAME), "w") as fi:
            json.dump(self.serializable_attrs, fi)
        with open(os.path.join(filepath, self._UNIGRAMS_SERIALIZED_FILENAME), "w", encoding=unigram_encoding) as fi:
            for unigram in sorted(self._unigram_set):
                fi.write(unigram + "\n")
        src = self._kenlm_model.path.decode("utf-8")
        shutil.copy2(src, os.path.join(filepath, os.path.split(src)[1]))

    @staticmethod
    def parse_directory_contents(filepath: str) -> Dict[str, str]:
        """A saved model directory holds exactly three things (language_model.py:415-438): the attribute file, the unigram
        list and ONE n-gram model file (.arpa / .bin / .binary, or this package's flat .ctcdec)."""
        import os

        attrs_name, unigrams_name = LanguageModel._ATTRS_SERIALIZED_FILENAME, LanguageModel._UNIGRAMS_SERIALIZED_FILENAME
        entries = sorted(e for e in os.listdir(filepath) if not e.startswith((".", "__")))
        if len(entries) != 3:
            raise ValueError(f"Found wrong number of files in directory. Expected 3 files, found {entries}")
        if attrs_name not in entries:
            raise ValueError(f"did not find attributes file in files: {entries}")
        if unigrams_name not in entries:
            raise ValueError(f"did not find unigrams file in files: {entries}")
        (model_name,) = [e for e in entries if e not in (attrs_name, unigrams_name)]
        if os.path.splitext(model_name)[1] not in (".arpa", ".bin", ".binary", NgramModel.FLAT_SUFFIX):
            raise ValueError(f"Expected the n-gram model file to end in `.arpa` or `.bin(ary)`. Found {model_name}")
        return {"json_attrs": os.path.join(filepath, attrs_name), "unigrams": os.path.join(filepath, unigrams_name),
                "kenlm": os.path.join(filepath, model_name)}

    @classmethod
    def load_from_dir(cls, filepath: str, unigram_encoding: Optional[str] = None) -> "LanguageModel":
        """Inverse of save_to_dir (language_model.py:440-452): weights from the attribute file, one unigram per line."""
        import json

        parts = cls.parse_directory_contents(filepath)
        with open(parts["json_attrs"]) as src:
            weights = json.load(src)
        if sorted(weights) != sorted(cls.JSON_ATTRS):
            raise ValueError(f"Expected json serialized attributes to be {cls.JSON_ATTRS} but found {list(weights)}")
        with open(parts["unigrams"], encoding=unigram_encoding) as src:
            unigrams = src.read().splitlines()
        return cls(NgramModel(parts["kenlm"]), unigrams, **weights)

    def reset_params(self, **params: Dict[str, Any]) -> None:
        """language_model.py:271-301."""
        for name, typ, attr in (
            ("alpha", float, "alpha"),
            ("beta", float, "beta"),
            ("unk_score_offset", float, "unk_score_offset"),
            ("score_boundary", bool, "score_boundary"),
        ):
            val = params.get(name)
            if val is not None:
                if not isinstance(val, typ):
                    raise ValueError(f"{name} must be a {typ.__name__}. Got {type(val)}.")
                setattr(self, attr, val)

    @property
    def order(self) -> int:
        return self._kenlm_model.order

    def get_start_state(self) -> KenlmState:
        return KenlmState(self._kenlm_model.start_state(self.score_boundary))

    def _get_raw_end_score(self, start_state: NgramState) -> float:
        if self.score_boundary:
            return self._kenlm_model.BaseScore(start_state, "</s>", NgramState())
        return 0.0

    def score_partial_token(self, partial_token: str) -> float:
        if not self._has_trie:
            is_oov = 1.0
        else:
            is_oov = int((self._kenlm_model.prefix_flags(partial_token) & 1) == 0) if partial_token else int(
                self._n_unigrams == 0)
        unk_score = self.unk_score_offset * is_oov
        if len(partial_token) > AVG_TOKEN_LEN:
            unk_score = unk_score * len(partial_token) / AVG_TOKEN_LEN
        return unk_score

    def score(
        self, prev_state: AbstractLMState, word: str, is_last_word: bool = False
    ) -> Tuple[float, KenlmState]:
        if not isinstance(prev_state, KenlmState):
            raise AssertionError(f"Wrong input state type found. Expected KenlmState, got {type(prev_state)}")
        end_state = NgramState()
        lm_score = self._kenlm_model.BaseScore(prev_state.state, word, end_state)
        flags = self._kenlm_model.prefix_flags(word)
        in_unigrams = bool(flags & 4)
        in_model = word in self._kenlm_model
        if (self._n_unigrams > 0 and not in_unigrams) or not in_model:
            lm_score += self.unk_score_offset
        if is_last_word:
            lm_score = lm_score + self._get_raw_end_score(end_state)
        lm_score = self.alpha * lm_score * LOG_BASE_CHANGE_FACTOR + self.beta
        return lm_score, KenlmState(end_state)


MAX_LANGUAGE_MODELS = 4  # CTCDEC_MAX_LMS


class MultiLanguageModel(AbstractLanguageModel):
    """language_model.py:455-502: several n-gram LanguageModels scored side by side -- word scores
    averaged, partial-word scores averaged, history pruning by the largest order.  The decode path runs
    the same arithmetic on the device (ctcdec_lm_share_multi); the methods here serve the public API."""

    def __init__(self, language_models: Sequence[AbstractLanguageModel]) -> None:
        if len(language_models) < 2:
            raise ValueError("This class is meant to contain at least 2 language models.")
        self._language_models = list(language_models)

    @property
    def language_models(self) -> List[AbstractLanguageModel]:
        return list(self._language_models)

    @property
    def order(self) -> int:
        return max([lm.order for lm in self._language_models])

    def get_start_state(self) -> MultiLanguageModelState:
        return MultiLanguageModelState([lm.get_start_state() for lm in self._language_models])

    def score_partial_token(self, partial_token: str) -> float:
        scores = [lm.score_partial_token(partial_token) for lm in self._language_models]
        total = 0.0
        for sc in scores:
            total += sc
        return float(total / len(scores))

    def score(
        self, prev_state: AbstractLMState, word: str, is_last_word: bool = False
    ) -> Tuple[float, MultiLanguageModelState]:
        """Every member scores `word` from its own state; the scores are averaged (left-to-right sum, then one
        division, as in the reference: language_model.py:495-501)."""
        members = self._language_models
        if not isinstance(prev_state, MultiLanguageModelState):
            raise AssertionError(
                "Wrong input state type found. Expected MultiLanguageModelState, got %s" % type(prev_state))
        if len(prev_state.states) != len(members):
            raise AssertionError("Number of states (%d) does not match number of language models (%d)." % (
                len(prev_state.states), len(members)))
        scored = [lm.score(st, word, is_last_word=is_last_word) for st, lm in zip(prev_state.states, members)]
        total = 0.0
        for value, _ in scored:
            total += value
        return total / len(members), MultiLanguageModelState([st for _, st in scored])
