"""Synthetic vocabularies, n-gram LMs and logit generators shared by bench.py, tests/ and
oracle/make_golden.py.  Nothing here is on the product
path; it only manufactures inputs.

All generators are seeded: utterance ``u`` of config ``c`` uses
``np.random.default_rng(1_000_003*c + u)``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# reference tests/test_decoder.py:155-184 (28 labels; the alphabet appends the blank => V=29)
LIBRI_LABELS = [" "] + list("abcdefghijklmnopqrstuvwxyz") + ["'"]
# tutorials/02_pipeline_huggingface.ipynb:87 (HF Wav2Vec2-base vocab, char level, V=32)
HF_W2V2_LABELS = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("ETAONIHSRDLUMWCFGYPBVK'XJQZ")

_LETTERS = "abcdefghijklmnopqrstuvwxyz"
_LETTER_P = np.array(
    [8.2, 1.5, 2.8, 4.3, 12.7, 2.2, 2.0, 6.1, 7.0, 0.15, 0.8, 4.0, 2.4, 6.7, 7.5, 1.9, 0.1, 6.0,
     6.3, 9.1, 2.8, 1.0, 2.4, 0.15, 2.0, 0.07]
)
_LETTER_P = _LETTER_P / _LETTER_P.sum()


def make_words(n_words: int, seed: int = 7, upper: bool = False) -> List[str]:
    """``n_words`` distinct pseudo-words, lengths 1..11, english-like letter frequencies."""
    rng = np.random.default_rng(seed)
    words: List[str] = []
    seen = set()
    while len(words) < n_words:
        ln = int(min(11, max(1, rng.poisson(4.2) + 1)))
        w = "".join(rng.choice(list(_LETTERS), size=ln, p=_LETTER_P))
        if upper:
            w = w.upper()
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def make_sentences(words: Sequence[str], n_sent: int, seed: int = 11, max_len: int = 12) -> List[List[int]]:
    """Zipf-distributed word-id sentences with a little first-order structure."""
    rng = np.random.default_rng(seed)
    nw = len(words)
    ranks = np.arange(1, nw + 1, dtype=np.float64)
    p = 1.0 / ranks ** 1.05
    p /= p.sum()
    # each word has a few preferred successors so that higher-order n-grams recur
    succ = rng.integers(0, nw, size=(nw, 4))
    out = []
    for _ in range(n_sent):
        ln = int(rng.integers(3, max_len + 1))
        s = [int(rng.choice(nw, p=p))]
        for _k in range(ln - 1):
            if rng.random() < 0.6:
                s.append(int(succ[s[-1], rng.integers(0, 4)]))
            else:
                s.append(int(rng.choice(nw, p=p)))
        out.append(s)
    return out


def write_arpa(
    path: str,
    words: Sequence[str],
    sentences: Sequence[Sequence[int]],
    order: int = 4,
    seed: int = 13,
    max_ngrams: Optional[Dict[int, int]] = None,
    with_unk: bool = True,
) -> Dict[int, int]:
    """Write a well-formed ARPA file (every n-gram's prefix and suffix present) whose n-grams are
    those observed in ``sentences`` (with <s>/</s>), log10 probs ~U(-6,-0.1), backoffs ~U(-1.5,0)."""
    rng = np.random.default_rng(seed)
    toks = ["<s>", "</s>"] + list(words)
    grams: List[Dict[Tuple[int, ...], None]] = [dict() for _ in range(order + 1)]
    for w in range(len(toks)):
        grams[1][(w,)] = None
    for s in sentences:
        seq = [0] + [w + 2 for w in s] + [1]
        full = False
        for n in range(2, order + 1):
            cap = max_ngrams.get(n) if max_ngrams else None
            if cap is not None and len(grams[n]) >= cap:
                full = True
                break
        if full:
            break
        for n in range(2, order + 1):
            for i in range(len(seq) - n + 1):
                grams[n][tuple(seq[i : i + n])] = None
    counts = {n: len(grams[n]) + (1 if (n == 1 and with_unk) else 0) for n in range(1, order + 1)}
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\n")
        for n in range(1, order + 1):
            f.write("ngram %d=%d\n" % (n, counts[n]))
        for n in range(1, order + 1):
            f.write("\n\\%d-grams:\n" % n)
            keys = list(grams[n].keys())
            probs = rng.uniform(-6.0, -0.1, size=len(keys))
            bos = rng.uniform(-1.5, 0.0, size=len(keys))
            if n == 1 and with_unk:
                f.write("-7.5\t<unk>\t0\n")
            for k, pr, bo in zip(keys, probs, bos):
                ws = " ".join(toks[i] for i in k)
                ends = k[-1] == 1  # ...</s> cannot be extended
                if n == 1:
                    if ws == "<s>":
                        f.write("-99\t<s>\t%.6f\n" % bo)
                    else:
                        f.write("%.6f\t%s\t%s\n" % (pr, ws, "0" if ends else "%.6f" % bo))
                elif n == order or ends:
                    f.write("%.6f\t%s\n" % (pr, ws))
                else:
                    f.write("%.6f\t%s\t%.6f\n" % (pr, ws, bo))
        f.write("\n\\end\\\n")
    return counts


def make_bpe_vocab(words: Sequence[str], size: int = 1023, seed: int = 17) -> List[str]:
    """Sentencepiece-style piece list of ``size`` entries (the alphabet appends the blank):
    "<unk>", the bare boundary piece, single letters with and without the boundary mark, then the
    most frequent word-initial and word-internal substrings of ``words``."""
    mark = "▁"
    pieces: List[str] = ["<unk>", mark]
    pieces += list(_LETTERS)
    pieces += [mark + ch for ch in _LETTERS]
    seen = set(pieces)
    lead: Dict[str, float] = {}
    mid: Dict[str, float] = {}
    for r, w in enumerate(words):
        wt = 1.0 / (r + 1.0)
        for ln in range(2, 6):
            if len(w) >= ln:
                lead[w[:ln]] = lead.get(w[:ln], 0.0) + wt
            for i in range(1, len(w) - ln + 1):
                mid[w[i : i + ln]] = mid.get(w[i : i + ln], 0.0) + wt
    lead_sorted = sorted(lead.items(), key=lambda kv: (-kv[1], kv[0]))
    mid_sorted = sorted(mid.items(), key=lambda kv: (-kv[1], kv[0]))
    li = mi = 0
    while len(pieces) < size:
        if li < len(lead_sorted) and (li <= mi or mi >= len(mid_sorted)):
            p = mark + lead_sorted[li][0]
            li += 1
        elif mi < len(mid_sorted):
            p = mid_sorted[mi][0]
            mi += 1
        else:  # pragma: no cover - vocabulary exhausted
            p = "zz%d" % len(pieces)
        if p not in seen:
            seen.add(p)
            pieces.append(p)
    return pieces


def _tokenise_bpe(word: str, piece_ids: Dict[str, int], first: bool = True) -> List[int]:
    """Greedy longest-match segmentation of one word into pieces (word-initial piece carries the
    boundary mark)."""
    mark = "▁"
    out = []
    pos = 0
    while pos < len(word):
        for ln in range(min(6, len(word) - pos), 0, -1):
            sub = word[pos : pos + ln]
            key = (mark + sub) if pos == 0 else sub
            if key in piece_ids:
                out.append(piece_ids[key])
                pos += ln
                break
        else:  # pragma: no cover - single letters always exist
            raise ValueError("cannot tokenise %r" % word)
    return out


def words_to_path(
    sent: Sequence[str], labels: Sequence[str], is_bpe: bool, space_label: str = " "
) -> List[int]:
    """Token ids spelling ``sent`` with ``labels`` (raw, un-normalised label strings)."""
    ids = {lab: i for i, lab in enumerate(labels)}
    path: List[int] = []
    for k, w in enumerate(sent):
        if is_bpe:
            path += _tokenise_bpe(w, ids)
        else:
            if k > 0:
                path.append(ids[space_label])
            path += [ids[ch] for ch in w]
    return path


def ctc_stretch(path: Sequence[int], T: int, blank: int, rng: np.random.Generator) -> np.ndarray:
    """Frame-level alignment of length T: each token held 1-3 frames, a blank between tokens with
    probability 0.5 (always between equal neighbours), padded with blanks / truncated to T."""
    frames: List[int] = []
    prev = -1
    for tok in path:
        if tok == prev or rng.random() < 0.5:
            frames.append(blank)
        frames += [tok] * int(rng.integers(1, 4))
        prev = tok
        if len(frames) >= T:
            break
    frames = frames[:T]
    frames += [blank] * (T - len(frames))
    return np.asarray(frames, dtype=np.int64)


def ctc_stretch_peaky(path: Sequence[int], T: int, blank: int, rng: np.random.Generator, hold: float = 0.35,
                      gap: float = 2.2) -> np.ndarray:
    """Frame-level alignment shaped like a trained CTC model's: a token is one frame (held a second / third one
    with probability `hold` each), tokens are separated by blank runs of geometric length (mean `gap`)."""
    frames: List[int] = []
    for tok in path:
        frames += [blank] * int(rng.geometric(1.0 / (1.0 + gap)) - 1 + (1 if frames and frames[-1] == tok else 0))
        frames.append(tok)
        while rng.random() < hold and len(frames) % 7:
            frames.append(tok)
        if len(frames) >= T:
            break
    frames = frames[:T]
    frames += [blank] * (T - len(frames))
    return np.asarray(frames, dtype=np.int64)


def d_peaky(
    config: int,
    utt: int,
    T: int,
    labels: Sequence[str],
    is_bpe: bool,
    words: Sequence[str],
    sentences: Sequence[Sequence[int]],
    blank: int,
    boost: float = 16.0,
    unsure: float = 0.12,
    space_label: str = " ",
) -> np.ndarray:
    """Real-posterior-like case (statistics of the reference's tests/sample_data/libri_logits.json: 327 of 371
    frames have ONE label above token_min_logp, 43 % of all frames only the blank, blank-only runs of 3.3 frames on
    average, the other 12 % of the frames carry one or two competitors 0.02-4 nats below the best): a confident
    one-hot of a peaky alignment, plus competitors on a fraction `unsure` of the frames."""
    rng = np.random.default_rng(1_000_003 * config + utt)
    V = len(labels) if blank < len(labels) else len(labels) + 1
    path: List[int] = []
    while len(path) * 3 < T:
        s = sentences[int(rng.integers(0, len(sentences)))]
        p = words_to_path([words[i] for i in s], labels, is_bpe, space_label)
        if path and not is_bpe:
            path.append(labels.index(space_label))
        path += p
    ali = ctc_stretch_peaky(path, T, blank, rng)
    x = rng.standard_normal((T, V)).astype(np.float32)
    x[np.arange(T), ali] += np.float32(boost)
    soft = np.nonzero(rng.random(T) < unsure)[0]
    for t in soft:
        for _ in range(1 if rng.random() < 0.9 else 2):
            x[t, int(rng.integers(0, V))] = x[t, ali[t]] - np.float32(rng.uniform(0.0, 4.0))
    return x


def d_flat(config: int, utt: int, T: int, V: int) -> np.ndarray:
    """Stress case: raw N(0,1) logits (the reference's own fuzz generator)."""
    rng = np.random.default_rng(1_000_003 * config + utt)
    return rng.standard_normal((T, V)).astype(np.float32)


def d_words(
    config: int,
    utt: int,
    T: int,
    labels: Sequence[str],
    is_bpe: bool,
    words: Sequence[str],
    sentences: Sequence[Sequence[int]],
    blank: int,
    boost: float = 6.0,
    space_label: str = " ",
) -> np.ndarray:
    """Headline case: noisy one-hot of a CTC-stretched spelling of LM sentences."""
    rng = np.random.default_rng(1_000_003 * config + utt)
    V = len(labels) if blank < len(labels) else len(labels) + 1
    sent: List[str] = []
    path: List[int] = []
    while len(path) * 2 < T:
        s = sentences[int(rng.integers(0, len(sentences)))]
        sent = [words[i] for i in s]
        p = words_to_path(sent, labels, is_bpe, space_label)
        if path and not is_bpe:
            path.append(labels.index(space_label))
        path += p
    ali = ctc_stretch(path, T, blank, rng)
    x = rng.standard_normal((T, V)).astype(np.float32)
    x[np.arange(T), ali] += np.float32(boost)
    return x


class SynthLM:
    """A seeded synthetic LM bundle on disk: words, sentences and the ARPA file."""

    def __init__(self, directory: str, n_words: int, n_sent: int, order: int = 4, seed: int = 7,
                 upper: bool = False, max_ngrams: Optional[Dict[int, int]] = None):
        os.makedirs(directory, exist_ok=True)
        self.words = make_words(n_words, seed=seed, upper=upper)
        self.sentences = make_sentences(self.words, n_sent, seed=seed + 4)
        self.path = os.path.join(directory, "synth_%dw_%ds_o%d_s%d%s.arpa" % (
            n_words, n_sent, order, seed, "u" if upper else ""))
        if not os.path.exists(self.path):
            tmp = self.path + ".tmp%d" % os.getpid()
            self.counts = write_arpa(tmp, self.words, self.sentences, order=order, seed=seed + 6,
                                     max_ngrams=max_ngrams)
            os.replace(tmp, self.path)
        self.order = order

    def hotwords(self, n_in: int = 20, n_oov: int = 5, seed: int = 23) -> List[str]:
        rng = np.random.default_rng(seed)
        pick = [self.words[int(i)] for i in rng.choice(min(len(self.words), 2000), size=n_in, replace=False)]
        oov = []
        while len(oov) < n_oov:
            w = "".join(rng.choice(list(_LETTERS), size=int(rng.integers(5, 9))))
            if w not in self.words and w not in oov:
                oov.append(w)
        return pick + oov
