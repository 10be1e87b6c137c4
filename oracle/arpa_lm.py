"""ORACLE (test infrastructure, never shipped): ARPA back-off n-gram scorer in plain Python.

This restates the query algorithm of the *external* dependency the reference binds to:
kenlm (github.com/kpu/kenlm, version UNPINNED by the reference -- CI installs
``archive/master.zip``, see /root/reference/.github/workflows/tests_and_lint.yml:32,51; the
package is absent from /root/reference/pyproject.toml:18-22 and its source is NOT under
/root/reference).  What is restated is kenlm's published behaviour for the calls the reference
makes (call sites /root/reference/pyctcdecode/language_model.py:95,109,306,312,314,321,347,352):

* ``Model(path)``: ARPA text -> vocabulary (``<unk>`` is index 0; a missing ``<unk>`` gets
  log10 p = -100, backoff 0), probabilities and back-offs stored as fp32.
* ``BaseScore(in_state, word, out_state)``: longest-match log10 probability of ``word`` given the
  context words held by ``in_state`` plus the back-off weights of every skipped context,
  accumulated in fp32, shortest context first (kenlm ``GenericModel::FullScore``).
* ``BeginSentenceWrite`` (context ``<s>``), ``NullContextWrite`` (empty context),
  ``word in model`` (vocabulary index != 0, so "<unk>" itself is "not in"), ``order``, ``path``.

PARITY STATUS: pinned only through the reference's own kenlm-boundary tests (toy 2-gram
``bugs_bunny_kenlm.arpa``; /root/reference/pyctcdecode/tests/test_decoder.py:245-295,324-384,
426-513,560-584).  For >=3-gram back-off chains, binary formats and quantisation the real kenlm is
not available in this environment: "parity unpinned" against kenlm itself.

State convention (equivalent to kenlm's minimised state for well-formed ARPA files): the state
after a word holds the words of the longest matched n-gram (newest first, at most order-1) and the
back-off weight of every context length it holds (0.0 where ARPA omits the column).
"""
from __future__ import annotations

import numpy as np

_F32 = np.float32


class ArpaState:
    """Context words (newest first) + their back-off weights. Mirrors kenlm.State."""

    __slots__ = ("words", "backoff")

    def __init__(self):
        self.words = ()  # tuple of word ids, newest first
        self.backoff = ()  # tuple of np.float32, backoff[k] = backoff of newest k+1 words

    def copy_from(self, other: "ArpaState") -> None:
        self.words = other.words
        self.backoff = other.backoff

    def __eq__(self, other):
        return (
            isinstance(other, ArpaState)
            and self.words == other.words
            and tuple(float(b) for b in self.backoff) == tuple(float(b) for b in other.backoff)
        )

    def __hash__(self):
        return hash((self.words, tuple(float(b) for b in self.backoff)))


class ArpaModel:
    """ARPA n-gram model with kenlm query semantics (fp32 storage and accumulation)."""

    def __init__(self, path: str):
        self.path = path.encode("utf-8") if isinstance(path, str) else path
        self.vocab = {"<unk>": 0}
        self.words = ["<unk>"]
        # ngrams[n] maps tuple(ids oldest..newest) -> (prob f32, backoff f32)
        self.ngrams = {}
        self.order = 0
        self._load(path if isinstance(path, str) else path.decode("utf-8"))

    # -- loading -------------------------------------------------------------------------------
    def _load(self, path: str) -> None:
        counts = {}
        section = 0
        unk_seen = False
        with open(path, "r", encoding="utf-8") as f:
            for raw in f:
                line = raw.strip()
                if not line:
                    continue
                if line.startswith("\\"):
                    if line == "\\data\\":
                        section = 0
                    elif line == "\\end\\":
                        break
                    elif line.endswith("-grams:"):
                        section = int(line[1 : line.index("-")])
                        self.ngrams.setdefault(section, {})
                    continue
                if section == 0:
                    if line.startswith("ngram "):
                        n, c = line[6:].split("=")
                        counts[int(n)] = int(c)
                    continue
                fields = line.split("\t") if "\t" in line else line.split()
                if "\t" in line:
                    prob = _F32(fields[0])
                    toks = fields[1].split(" ")
                    backoff = _F32(fields[2]) if len(fields) > 2 and fields[2] != "" else _F32(0.0)
                else:
                    prob = _F32(fields[0])
                    toks = fields[1 : 1 + section]
                    backoff = _F32(fields[1 + section]) if len(fields) > 1 + section else _F32(0.0)
                if len(toks) != section:
                    raise ValueError("malformed ARPA line in %d-gram section: %r" % (section, raw))
                if prob > 0:
                    raise ValueError("positive log probability in ARPA: %r" % raw)
                if section == 1:
                    w = toks[0]
                    if w == "<unk>":
                        unk_seen = True
                        idx = 0
                    else:
                        idx = self.vocab.get(w)
                        if idx is None:
                            idx = len(self.words)
                            self.vocab[w] = idx
                            self.words.append(w)
                    self.ngrams[1][(idx,)] = (prob, backoff)
                else:
                    ids = tuple(self.vocab.get(t, 0) for t in toks)
                    self.ngrams[section][ids] = (prob, backoff)
        self.order = max(counts) if counts else max(self.ngrams)
        if not unk_seen:
            self.ngrams.setdefault(1, {})[(0,)] = (_F32(-100.0), _F32(0.0))

    # -- kenlm.Model API -----------------------------------------------------------------------
    def index(self, word: str) -> int:
        return self.vocab.get(word, 0) if word != "<unk>" else 0

    def __contains__(self, word: str) -> bool:
        return self.index(word) != 0

    def BeginSentenceWrite(self, state: ArpaState) -> None:  # noqa: N802 (kenlm name)
        bos = self.index("<s>")
        state.words = (bos,)
        state.backoff = (self.ngrams[1].get((bos,), (_F32(0), _F32(0)))[1],)
        if self.order < 2:
            state.words, state.backoff = (), ()

    def NullContextWrite(self, state: ArpaState) -> None:  # noqa: N802
        state.words = ()
        state.backoff = ()

    def BaseScore(self, in_state: ArpaState, word: str, out_state: ArpaState) -> float:  # noqa: N802
        return self.base_score_id(in_state, self.index(word), out_state)

    def base_score_id(self, in_state: ArpaState, wid: int, out_state: ArpaState) -> float:
        prob, bo = self.ngrams[1][(wid,)]
        out_bo = [bo]
        matched = 1
        ctx = in_state.words  # newest first
        for n in range(2, min(self.order, len(ctx) + 1) + 1):
            key = tuple(reversed(ctx[: n - 1])) + (wid,)
            hit = self.ngrams.get(n, {}).get(key)
            if hit is None:
                break
            prob, bo = hit
            out_bo.append(bo)
            matched = n
        total = _F32(prob)
        for i in range(matched - 1, len(ctx)):
            total = _F32(total + in_state.backoff[i])
        keep = min(matched, self.order - 1)
        new_words = ((wid,) + tuple(ctx))[:keep]
        new_bo = tuple(out_bo[:keep])
        out_state.words = new_words
        out_state.backoff = new_bo
        return float(total)
