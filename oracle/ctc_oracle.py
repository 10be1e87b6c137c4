"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the reference's CTC prefix-beam-search hot path, written from its semantics
(SURVEY.md App. A/B/G), used ONLY by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg as the checker / reported baseline.  The product (pyctcdecode_amd) never
imports this file.

Reference functions restated (all under /root/reference/pyctcdecode/):
  * input normalisation ............ decoder.py:759-765, _log_softmax decoder.py:180-197
  * token prune + iteration order .. decoder.py:444-447 (CPython ``set`` slot order; emulated by
                                     :func:`cpython_set_order`, Objects/setobject.c of CPython 3.10)
  * blank/repeat/boundary/append ... decoder.py:447-534
  * merge (logsumexp, donor rules) . decoder.py:170-177,200-224
  * LM / hot-word scoring .......... decoder.py:346-424, language_model.py:137-150,326-360
  * threshold prune, top-B ......... decoder.py:545-548,165-167
  * history prune .................. decoder.py:227-258
  * finalisation + output .......... decoder.py:558-602,653-667
  * batch shell .................... decoder.py:146-157,801-945
The n-gram arithmetic itself lives in oracle/arpa_lm.py (restatement of the external kenlm).

PARITY STATUS: pinned.  oracle/make_golden.py imports the unmodified reference (through the
stand-ins in oracle/refshim/) and writes tests/golden/*.json; tests/test_oracle_golden.py checks
this restatement against those vectors (beams, frames, order exact; scores to 1e-9) and against
the reference's own known-answer floats (tests/test_decoder.py:330-336,515-558 of the reference).
The LM arithmetic is pinned only as far as the reference pins kenlm (toy 2-gram): see arpa_lm.py.

Structure differs from the reference on purpose (it is also the executable spec of the device
algorithm): candidates carry explicit arrival indices, merging is a single ordered dict keyed on
(text, partial, last_char), LM/hot-word scores are pure functions with memo tables.
"""
from __future__ import annotations

import math
import multiprocessing as mp
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

try:  # allow both "import oracle.ctc_oracle" and running from inside oracle/
    from .arpa_lm import ArpaModel, ArpaState
except ImportError:  # pragma: no cover
    from arpa_lm import ArpaModel, ArpaState

BPE_MARK = "▁"  # alphabet.py:9
AVG_TOKEN_LEN = 6  # constants.py:16
MIN_TOKEN_CLIP_P = 1e-15  # constants.py:17
LOG_BASE_CHANGE_FACTOR = 1.0 / math.log10(math.e)  # constants.py:18
NULL_FRAMES = (-1, -1)


# --------------------------------------------------------------------------------------------
# CPython 3.10 set iteration order for  set(ascending ids) | {argmax}   (decoder.py:445)
# --------------------------------------------------------------------------------------------
class _SetTable:
    """Open-addressing table of CPython's setobject.c (LINEAR_PROBES=9, PERTURB_SHIFT=5)."""

    __slots__ = ("slots", "mask", "used")

    def __init__(self, size: int = 8):
        self.slots = [-1] * size
        self.mask = size - 1
        self.used = 0

    def insert_clean(self, key: int) -> None:
        mask = self.mask
        slots = self.slots
        perturb = key
        i = key & mask
        while True:
            if slots[i] < 0:
                slots[i] = key
                return
            if i + 9 <= mask:
                for j in range(1, 10):
                    if slots[i + j] < 0:
                        slots[i + j] = key
                        return
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & mask

    def resize(self, minused: int) -> None:
        newsize = 8
        while newsize <= minused:
            newsize <<= 1
        old = self.slots
        self.slots = [-1] * newsize
        self.mask = newsize - 1
        for k in old:
            if k >= 0:
                self.insert_clean(k)

    def add(self, key: int) -> None:
        mask = self.mask
        slots = self.slots
        perturb = key
        i = key & mask
        while True:
            probes = 9 if i + 9 <= mask else 0
            e = i
            while True:
                if slots[e] < 0:
                    slots[e] = key
                    self.used += 1
                    if self.used * 5 >= mask * 3:
                        self.resize(self.used * 2 if self.used > 50000 else self.used * 4)
                    return
                if slots[e] == key:
                    return
                e += 1
                if probes == 0:
                    break
                probes -= 1
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & mask


def cpython_set_order(ascending_ids: Sequence[int], argmax: int) -> List[int]:
    """Iteration order of ``set(ascending_ids) | {argmax}`` under CPython 3.10 (hash(int)=int)."""
    a = _SetTable()
    for k in ascending_ids:
        a.add(int(k))
    # A | B  ==  copy(A) then merge(B): set_copy -> set_merge into an empty set
    r = _SetTable()
    if a.used:
        if a.used * 5 >= r.mask * 3:
            r.resize(a.used * 2)
        if r.mask == a.mask:
            r.slots = list(a.slots)
        else:
            for k in a.slots:
                if k >= 0:
                    r.insert_clean(k)
        r.used = a.used
    # merge {argmax}: one big resize is decided BEFORE the membership test
    if (r.used + 1) * 5 >= r.mask * 3:
        r.resize((r.used + 1) * 2)
    r.add(int(argmax))
    return [k for k in r.slots if k >= 0]


# --------------------------------------------------------------------------------------------
# scorers
# --------------------------------------------------------------------------------------------
class HotwordOracle:
    """language_model.py:115-189 as set + prefix table (no regex, no trie)."""

    def __init__(self, hotwords: Optional[Iterable[str]], weight: float):
        words = [s.strip() for s in (hotwords or []) if len(s.strip()) > 0]
        unigrams: List[str] = []
        for phrase in words:
            unigrams.extend(phrase.split())
        self.weight = weight
        self.words = set(unigrams)
        self.min_len: Dict[str, int] = {}
        for w in unigrams:
            for k in range(len(w) + 1):
                p = w[:k]
                if p not in self.min_len or len(w) < self.min_len[p]:
                    self.min_len[p] = len(w)

    def __contains__(self, partial: str) -> bool:
        return partial in self.min_len

    def count(self, text: str) -> int:
        return sum(1 for w in text.split(" ") if w in self.words) if text else 0

    def score(self, text: str) -> float:
        return self.weight * self.count(text)

    def score_partial(self, partial: str) -> float:
        if partial in self.min_len:
            return self.weight * len(partial) / self.min_len[partial]
        return 0.0


class LMOracle:
    """language_model.py:230-360 on top of oracle/arpa_lm.py."""

    def __init__(
        self,
        model: ArpaModel,
        unigrams: Optional[Iterable[str]],
        alpha: float = 0.5,
        beta: float = 1.5,
        unk_score_offset: float = -10.0,
        score_boundary: bool = True,
    ):
        self.model = model
        if unigrams is None:
            self.unigram_set = set()
            self.prefixes = None
        else:
            self.unigram_set = set(w for w in set(unigrams) if w in model)
            self.prefixes = set()
            for w in self.unigram_set:
                for k in range(len(w) + 1):
                    self.prefixes.add(w[:k])
        self.alpha = alpha
        self.beta = beta
        self.unk_score_offset = unk_score_offset
        self.score_boundary = score_boundary

    @property
    def order(self) -> int:
        return self.model.order

    def start_state(self) -> ArpaState:
        st = ArpaState()
        if self.score_boundary:
            self.model.BeginSentenceWrite(st)
        else:
            self.model.NullContextWrite(st)
        return st

    def score_partial(self, partial: str) -> float:
        if self.prefixes is None:
            is_oov = 1.0
        else:
            is_oov = int(partial not in self.prefixes)
        s = self.unk_score_offset * is_oov
        if len(partial) > AVG_TOKEN_LEN:
            s = s * len(partial) / AVG_TOKEN_LEN
        return s

    def score(self, prev: ArpaState, word: str, is_last_word: bool) -> Tuple[float, ArpaState]:
        end = ArpaState()
        lm = self.model.BaseScore(prev, word, end)
        if (len(self.unigram_set) > 0 and word not in self.unigram_set) or word not in self.model:
            lm += self.unk_score_offset
        if is_last_word:
            if self.score_boundary:
                lm = lm + self.model.BaseScore(end, "</s>", ArpaState())
            else:
                lm = lm + 0.0
        return self.alpha * lm * LOG_BASE_CHANGE_FACTOR + self.beta, end


class MultiLMOracle:
    """language_model.py:455-502: several LMOracles side by side; a state is the list of their states."""

    def __init__(self, lms: Sequence[LMOracle]):
        if len(lms) < 2:
            raise ValueError("This class is meant to contain at least 2 language models.")
        self.lms = list(lms)

    @property
    def order(self) -> int:
        return max(lm.order for lm in self.lms)  # language_model.py:467-469

    def start_state(self) -> List[ArpaState]:
        return [lm.start_state() for lm in self.lms]

    def score_partial(self, partial: str) -> float:
        return float(np.mean([lm.score_partial(partial) for lm in self.lms]))  # language_model.py:477-481

    def score(self, prev: List[ArpaState], word: str, is_last_word: bool) -> Tuple[float, List[ArpaState]]:
        score = 0.0
        end = []
        for st, lm in zip(prev, self.lms):  # language_model.py:495-501
            sc, e = lm.score(st, word, is_last_word)
            score += sc
            end.append(e)
        return score / len(self.lms), end


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _lse2(a: float, b: float) -> float:
    if a >= b:
        return a + math.log(1 + math.exp(b - a))
    return b + math.log(1 + math.exp(a - b))


def _join(text: str, word: str) -> str:
    if not word:
        return text
    if not text:
        return word
    return text + " " + word


def normalise_logits(logits: np.ndarray, sniff_on: Optional[np.ndarray] = None) -> np.ndarray:
    """decoder.py:759-765 / :180-197.

    ``sniff_on`` (tests only): the parity tests hand the oracle an exact float64 upcast of the float32 / float16 matrix
    the product gets, so that the reference's own low-precision log-softmax does not blur the comparison. The
    "probabilities or logits?" test, however, is a property of the ORIGINAL dtype (its mean row sum rounds to exactly 1
    or it does not): it is taken on ``sniff_on`` when given."""
    probe = logits if sniff_on is None else sniff_on
    if math.isclose(probe.sum(axis=1).mean(), 1):
        return np.log(np.clip(logits, MIN_TOKEN_CLIP_P, 1))
    x_max = np.amax(logits, axis=1, keepdims=True)
    x_max[~np.isfinite(x_max)] = 0
    tmp = logits - x_max
    with np.errstate(divide="ignore"):
        lse = np.log(np.sum(np.exp(tmp), axis=1, keepdims=True))
    return np.clip(tmp - lse, np.log(MIN_TOKEN_CLIP_P), 0)


class OBeam:
    """A live prefix.  Field meaning = reference ``Beam`` (decoder.py:69-79); next_word is
    always folded into text between frames so it is not stored."""

    __slots__ = ("text", "partial", "last", "tframes", "pframes", "logit", "lm")

    def __init__(self, text, partial, last, tframes, pframes, logit, lm=None):
        self.text = text
        self.partial = partial
        self.last = last
        self.tframes = tframes
        self.pframes = pframes
        self.logit = logit
        self.lm = lm


class OracleState:
    """Streaming state: live beams + memo tables (decoder.py:669-679)."""

    def __init__(self, beams, text_memo, partial_memo):
        self.beams = beams
        self.text_memo = text_memo
        self.partial_memo = partial_memo


# Decoders live in a module-level registry so that fork-pool workers find them (and their LM) in
# inherited memory instead of un-pickling them per task -- the reference does the same with its
# class-level ``model_container`` (decoder.py:262-269, 289-290).
_REGISTRY: Dict[int, "OracleDecoder"] = {}


def _decode_by_key(key: int, kw: dict, logits: np.ndarray) -> str:
    return _REGISTRY[key].decode(logits, **kw)


class OracleDecoder:
    """Restated BeamSearchDecoderCTC.  ``labels`` must already be normalised
    (alphabet.py:34-110): blank is "", word separator is " " (char) or a leading U+2581 (BPE)."""

    def __init__(self, labels: Sequence[str], is_bpe: bool, lm: Optional[LMOracle] = None):
        self.labels = list(labels)
        self.is_bpe = is_bpe
        self.lm = lm
        self._key = id(self)
        _REGISTRY[self._key] = self

    # -- scoring (pure functions with memo) ----------------------------------------------------
    def _text_entry(self, memo, text: str, word: str, hw: HotwordOracle, eos: bool):
        """memo[(text (+) word, eos)] -> (lm+hotword, raw lm, state); decoder.py:386-396."""
        new_text = _join(text, word)
        key = (new_text, eos)
        hit = memo.get(key)
        if hit is None:
            _, prev_raw, prev_state = memo[(text, False)]
            s, end = self.lm.score(prev_state, word, eos)
            raw = prev_raw + s
            hit = (raw + hw.score(new_text), raw, end)
            memo[key] = hit
        return hit

    def _score(self, cand: OBeam, word: str, hw: HotwordOracle, st: OracleState, eos: bool):
        """Fold ``word`` into cand.text and set cand.lm (decoder.py:346-424)."""
        new_text = _join(cand.text, word)
        if self.lm is None:
            cand.lm = cand.logit + hw.score(new_text) + hw.score_partial(cand.partial)
        else:
            lm_hw = self._text_entry(st.text_memo, cand.text, word, hw, eos)[0]
            part = cand.partial
            if len(part) > 0:
                ps = st.partial_memo.get(part)
                if ps is None:
                    ps = hw.score_partial(part) if part in hw else self.lm.score_partial(part)
                    st.partial_memo[part] = ps
                lm_hw += ps
            cand.lm = cand.logit + lm_hw
        cand.text = new_text

    @staticmethod
    def _select(cands: List[OBeam], beam_width: int, prune_logp: float) -> List[OBeam]:
        best = max(c.lm for c in cands)
        kept = [c for c in cands if c.lm >= best + prune_logp]
        # stable descending sort: exact ties keep arrival order (heapq.nlargest semantics)
        kept.sort(key=lambda c: -c.lm)
        return kept[:beam_width]

    # -- the per-frame recursion ---------------------------------------------------------------
    def _advance(
        self,
        logp: np.ndarray,
        st: OracleState,
        beam_width: int,
        prune_logp: float,
        token_min_logp: float,
        prune_history: bool,
        hw: HotwordOracle,
        first_frame: int,
    ) -> None:
        labels = self.labels
        is_bpe = self.is_bpe
        force_break = False  # one flag for the whole call (decoder.py:442)
        beams = st.beams
        for t in range(logp.shape[0]):
            row = logp[t]
            frame = first_frame + t
            amax = int(row.argmax())
            order = cpython_set_order(np.nonzero(row >= token_min_logp)[0], amax)
            merged: Dict[Tuple[str, str, str], Tuple[OBeam, str]] = {}
            for c in order:
                p = row[c]
                ch = labels[c]
                lead = is_bpe and ch[:1] == BPE_MARK
                for b in beams:
                    word = ""
                    if ch == "" or b.last == ch:
                        pf = b.pframes if ch == "" else (b.pframes[0], frame + 1)
                        cand = OBeam(b.text, b.partial, ch, b.tframes, pf, b.logit + p)
                    elif is_bpe and (lead or force_break):
                        force_break = False
                        clean = ch[1:] if lead else ch
                        if ch[-1:] == BPE_MARK:
                            clean = clean[:-1]
                            force_break = True
                        tf = b.tframes if b.partial == "" else b.tframes + (b.pframes,)
                        word = b.partial
                        cand = OBeam(b.text, clean, ch, tf, (frame, frame + 1), b.logit + p)
                    elif (not is_bpe) and ch == " ":
                        tf = b.tframes if b.partial == "" else b.tframes + (b.pframes,)
                        word = b.partial
                        cand = OBeam(b.text, "", ch, tf, NULL_FRAMES, b.logit + p)
                    else:
                        start = frame if b.pframes[0] < 0 else b.pframes[0]
                        cand = OBeam(
                            b.text, b.partial + ch, ch, b.tframes, (start, frame + 1), b.logit + p
                        )
                    key = (_join(cand.text, word), cand.partial, ch)
                    seen = merged.get(key)
                    if seen is not None:
                        cand.logit = _lse2(seen[0].logit, cand.logit)
                    merged[key] = (cand, word)  # position = first arrival, payload = latest
            cands = []
            for cand, word in merged.values():
                self._score(cand, word, hw, st, eos=False)
                cands.append(cand)
            top = self._select(cands, beam_width, prune_logp)
            if prune_history:
                n_hist = max(1, (1 if self.lm is None else self.lm.order) - 1)
                seen_keys = set()
                beams = []
                for b in top:
                    k = (tuple(b.text.split()[-n_hist:]), b.partial, b.last)
                    if k not in seen_keys:
                        seen_keys.add(k)
                        beams.append(b)
            else:
                beams = top
        st.beams = beams

    def _finalise(
        self,
        st: OracleState,
        beam_width: int,
        prune_logp: float,
        hw: HotwordOracle,
        force_next_word: bool,
        is_end: bool,
    ) -> List[OBeam]:
        """decoder.py:558-602."""
        if force_next_word or is_end:
            merged: Dict[Tuple[str, str, None], Tuple[OBeam, str]] = {}
            for b in st.beams:
                tf = b.tframes if b.partial == "" else b.tframes + (b.pframes,)
                cand = OBeam(b.text, "", None, tf, NULL_FRAMES, b.logit)
                key = (_join(b.text, b.partial), "", None)
                seen = merged.get(key)
                if seen is not None:
                    cand.logit = _lse2(seen[0].logit, cand.logit)
                merged[key] = (cand, b.partial)
            pairs = list(merged.values())
        else:
            pairs = [(OBeam(b.text, b.partial, b.last, b.tframes, b.pframes, b.logit), "") for b in st.beams]
        cands = []
        for cand, word in pairs:
            self._score(cand, word, hw, st, eos=is_end)
            cands.append(cand)
        return self._select(cands, beam_width, prune_logp)

    # -- public surface (same argument meaning as the reference) -------------------------------
    def get_starting_state(self, lm_start_state: Optional[ArpaState] = None) -> OracleState:
        memo = {}
        if self.lm is not None:
            start = self.lm.start_state() if lm_start_state is None else lm_start_state
            memo[("", False)] = (0.0, 0.0, start)
        return OracleState([OBeam("", "", None, (), NULL_FRAMES, 0.0)], memo, {})

    def _check(self, logits: np.ndarray) -> None:
        if len(logits.shape) != 2:
            raise ValueError("Input logits have %s dimensions, but need 2: (time, vocabulary)" % len(logits.shape))
        if logits.shape[-1] != len(self.labels):
            raise ValueError(
                "Input logits shape is %s, but vocabulary is size %s. "
                "Need logits of shape: (time, vocabulary)" % (logits.shape, len(self.labels))
            )

    def partial_decode_beams(
        self,
        logits: np.ndarray,
        st: OracleState,
        processed_frames: int,
        beam_width: int = 100,
        beam_prune_logp: float = -10.0,
        token_min_logp: float = -5.0,
        prune_history: bool = False,
        hotwords: Optional[Iterable[str]] = None,
        hotword_weight: float = 10.0,
        force_next_word: bool = False,
        is_end: bool = False,
        sniff_on: Optional[np.ndarray] = None,
    ) -> List[OBeam]:
        """decoder.py:681-728; the caller feeds the returned beams back through ``st.beams``."""
        self._check(logits)
        hw = HotwordOracle(hotwords, hotword_weight) if hotwords is not None else HotwordOracle([], 0.0)
        self._advance(
            normalise_logits(logits, sniff_on), st, beam_width, beam_prune_logp, token_min_logp,
            prune_history, hw, processed_frames,
        )
        out = self._finalise(st, beam_width, beam_prune_logp, hw, force_next_word, is_end)
        st.beams = out
        return out

    def decode_beams(
        self,
        logits: np.ndarray,
        beam_width: int = 100,
        beam_prune_logp: float = -10.0,
        token_min_logp: float = -5.0,
        prune_history: bool = False,
        hotwords: Optional[Iterable[str]] = None,
        hotword_weight: float = 10.0,
        lm_start_state: Optional[ArpaState] = None,
        sniff_on: Optional[np.ndarray] = None,
    ):
        """Returns [(text, last_lm_state, [(word,(start,end))...], logit_score, lm_score)]."""
        self._check(logits)
        hw = HotwordOracle(hotwords, hotword_weight)
        st = self.get_starting_state(lm_start_state)
        self._advance(
            normalise_logits(logits, sniff_on), st, beam_width, beam_prune_logp, token_min_logp,
            prune_history, hw, 0,
        )
        final = self._finalise(st, beam_width, beam_prune_logp, hw, True, True)
        out = []
        for b in final:
            entry = st.text_memo.get((b.text, True))
            words = b.text.split()
            out.append(
                (
                    " ".join(words),
                    entry[2] if entry is not None else None,
                    list(zip(words, b.tframes)),
                    float(b.logit),
                    float(b.lm),
                )
            )
        return out

    def decode(self, logits: np.ndarray, **kw) -> str:
        kw["prune_history"] = True  # decoder.py:888
        return self.decode_beams(logits, **kw)[0][0]

    def decode_batch(self, pool, logits_list, **kw) -> List[str]:
        """decoder.py:895-945: fork pool => map, spawn pool or None => sequential."""
        if pool is not None and isinstance(pool._ctx, mp.context.SpawnContext):  # noqa: SLF001
            pool = None
        if pool is None:
            return [self.decode(x, **kw) for x in logits_list]
        import functools

        # workers were forked after this decoder was built: look it up by key (nothing big is pickled)
        return pool.map(functools.partial(_decode_by_key, self._key, kw), logits_list)


def build_oracle(
    labels: Sequence[str],
    is_bpe: bool,
    arpa_path: Optional[str] = None,
    unigrams: Optional[Iterable[str]] = None,
    alpha: float = 0.5,
    beta: float = 1.5,
    unk_score_offset: float = -10.0,
    lm_score_boundary: bool = True,
) -> OracleDecoder:
    """decoder.py:1051-1099 for already-normalised labels."""
    lm = None
    if arpa_path is not None:
        model = ArpaModel(arpa_path)
        if unigrams is None and arpa_path.endswith(".arpa"):
            unigrams = load_unigrams_from_arpa(arpa_path)
        lm = LMOracle(model, unigrams, alpha, beta, unk_score_offset, lm_score_boundary)
    return OracleDecoder(labels, is_bpe, lm)


def load_unigrams_from_arpa(path: str) -> set:
    """language_model.py:67-84: 1-gram lines with exactly three tab-separated fields."""
    out = set()
    on = False
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line == "\\1-grams:":
                on = True
            elif line == "\\2-grams:":
                break
            if on and line:
                parts = line.split("\t")
                if len(parts) == 3:
                    out.add(parts[1])
    if not out:
        raise ValueError("No unigrams found in arpa file. Something is wrong with the file.")
    return out
