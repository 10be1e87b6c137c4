"""Drop-in Python surface of the decoder (reference: pyctcdecode/decoder.py).

Same names, positional order and defaults as the reference for ``build_ctcdecoder``,
``BeamSearchDecoderCTC.decode / decode_beams / decode_batch / decode_beams_batch``, the frozen
dataclasses ``Beam / LMBeam / OutputBeam`` and ``reset_params / cleanup / clear_class_models``.
All decoding work happens in libctcdec (HIP, gfx950) through ctypes; this file only marshals.
The ``pool`` argument of the batch entry points is accepted for signature compatibility and
ignored: a batch is one device launch (one workgroup per utterance) instead of a process pool.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import logging
import math
import os
import threading
from typing import Any, Collection, Dict, Iterable, List, Optional, Sequence, Tuple, TypeVar

import numpy as np

from . import _binding as B
from .alphabet import Alphabet, verify_alphabet_coverage
from .constants import (
    DEFAULT_ALPHA,
    DEFAULT_BEAM_WIDTH,
    DEFAULT_BETA,
    DEFAULT_HOTWORD_WEIGHT,
    DEFAULT_MIN_TOKEN_LOGP,
    DEFAULT_PRUNE_BEAMS,
    DEFAULT_PRUNE_LOGP,
    DEFAULT_SCORE_LM_BOUNDARY,
    DEFAULT_UNK_LOGP_OFFSET,
    LOG_BASE_CHANGE_FACTOR,
)
from .language_model import (
    AbstractLanguageModel,
    AbstractLMState,
    HotwordScorer,
    KenlmState,
    LanguageModel,
    MAX_LANGUAGE_MODELS,
    MultiLanguageModel,
    MultiLanguageModelState,
    NgramModel,
    NgramState,
    _default_device,
    load_unigram_set_from_arpa,
)

logger = logging.getLogger(__name__)

Frames = Tuple[int, int]
WordFrames = Tuple[str, Frames]
# type aliases callers of the reference import from here (decoder.py:121-126, 142-143)
LMScoreCacheKey = Tuple[str, bool]
LMScoreCacheValue = Tuple[float, float, AbstractLMState]
LMScoreCache = Dict[LMScoreCacheKey, LMScoreCacheValue]
FloatVar = TypeVar("FloatVar", bound=np.floating)
Shape = TypeVar("Shape")


@dataclasses.dataclass(frozen=True)
class Beam:
    """decoder.py:69-92 (field order is API)."""

    text: str
    next_word: str
    partial_word: str
    last_char: Optional[str]
    text_frames: List[Frames]
    partial_frames: Frames
    logit_score: float

    @classmethod
    def from_lm_beam(cls, lm_beam: "LMBeam") -> "Beam":
        """The beam without its language-model score (decoder.py:81-92)."""
        return Beam(*(getattr(lm_beam, f.name) for f in dataclasses.fields(Beam)))


@dataclasses.dataclass(frozen=True)
class LMBeam(Beam):
    lm_score: float


@dataclasses.dataclass(frozen=True)
class OutputBeam:
    """decoder.py:102-118."""

    text: str
    last_lm_state: Optional[AbstractLMState]
    text_frames: List[WordFrames]
    logit_score: float
    lm_score: float

    def get_mp_safe_beam(self) -> "OutputBeam":
        """A copy that pickles: the LM state is swapped for its process-independent form (decoder.py:112-118)."""
        state = self.last_lm_state
        return dataclasses.replace(self, last_lm_state=state.get_mp_safe_state() if state is not None else None)


NULL_FRAMES: Frames = (-1, -1)
EMPTY_START_BEAM = Beam("", "", "", None, [], NULL_FRAMES, 0.0)


def _same_lm_state(a: Any, b: Any) -> bool:
    """Equality of two language-model states by value (one model's, or one per model of a MultiLanguageModel)."""
    if type(a) is not type(b):
        return False
    if hasattr(a, "states"):
        return len(a.states) == len(b.states) and all(_same_lm_state(x, y) for x, y in zip(a.states, b.states))
    return getattr(a, "state", a) == getattr(b, "state", b)


def _is_device_tensor(x: Any) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class _Batch:
    """Logit matrices of one call, normalised to what the C ABI takes (and shape-checked like
    _check_logits_dimension, decoder.py:330-344)."""

    def __init__(self, logits_list: Sequence[Any], n_labels: int):
        self.keep: List[Any] = []  # keeps buffers alive during the call
        self.ptrs: List[int] = []
        self.frames: List[int] = []
        self.is_device = False
        self.device_index: Optional[int] = None  # device of the logits when they are device tensors
        self.dtype = 0
        if getattr(logits_list, "ndim", 0) == 3 and self._from_3d(logits_list, n_labels):
            return
        n_dev = 0
        for x in logits_list:
            shape = x.shape
            if len(shape) != 2:
                raise ValueError("Input logits have %s dimensions, but need 2: (time, vocabulary)" % len(shape))
            if shape[1] != n_labels:
                raise ValueError(
                    "Input logits shape is %s, but vocabulary is size %s. "
                    "Need logits of shape: (time, vocabulary)" % (tuple(shape), n_labels)
                )
            if getattr(x, "is_cuda", False):
                n_dev += 1
        if n_dev not in (0, len(logits_list)):
            raise ValueError("cannot mix host arrays and device tensors in one batch")
        self.is_device = n_dev > 0
        keep, ptrs, frames = self.keep, self.ptrs, self.frames
        if self.is_device:
            import torch

            native = {torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.bfloat16: 3}
            dtypes = {x.dtype for x in logits_list}
            if len(dtypes) == 1 and next(iter(dtypes)) in native:
                target = next(iter(dtypes))  # fp16 / bf16 / fp32 / fp64 tensors are read in place
            else:
                target = torch.float64 if torch.float64 in dtypes or any(not d.is_floating_point for d in dtypes) \
                    else torch.float32
            for x in logits_list:
                t = x if x.dtype == target else x.to(target)
                if not t.is_contiguous():
                    t = t.contiguous()
                keep.append(t)
                ptrs.append(t.data_ptr())
                frames.append(t.shape[0])
            self.dtype = native[target]
            # Our kernels run on their own stream: the producer of the logits AND the dtype / layout conversions
            # enqueued just above (they run asynchronously on torch's stream) must be done before the native call
            # reads them -- so synchronise AFTER the conversions, on the stream of the tensors' own device.
            devs = {x.device for x in logits_list}
            if len(devs) != 1:
                raise ValueError("the logits of one batch must live on one device")
            dev = next(iter(devs))
            self.device_index = dev.index
            torch.cuda.current_stream(dev).synchronize()
        else:
            arrs = [x.detach().cpu().numpy() if _is_device_tensor(x) else np.asarray(x) for x in logits_list]
            # float32 / float16 matrices go over in their own dtype: the reference decides "probabilities or logits?" on
            # the input dtype (decoder.py:760) and computes a float32 matrix's log-softmax in float32 (decoder.py:180-197), and so
            # does the library. Everything else (float64, integers, batches that mix float32 with float64 / integer matrices --
            # those are computed in float64 altogether, a documented deviation of <= 1e-6 for their float32 members) is widened to
            # float64 first. Matrices without frames have no say.
            kinds = {a.dtype for a in arrs if a.shape[0] > 0}
            target, self.dtype = np.float64, 1
            if kinds == {np.dtype(np.float32)}:
                target, self.dtype = np.float32, 0
            elif kinds == {np.dtype(np.float16)}:
                target, self.dtype = np.float16, 2
            elif kinds and kinds <= {np.dtype(np.float32), np.dtype(np.float16)}:
                target, self.dtype = np.float32, 0
            for a in arrs:
                a = np.ascontiguousarray(a, dtype=target)
                keep.append(a)
                ptrs.append(a.ctypes.data if a.size else 0)
                frames.append(int(a.shape[0]))


    def _from_3d(self, batch: Any, n_labels: int) -> bool:
        """A padded [B, T, V] batch straight from the acoustic model: one pointer + strides instead of B
        per-utterance objects.  Returns False when the generic path has to take over."""
        if batch.shape[2] != n_labels:
            raise ValueError(
                "Input logits shape is %s, but vocabulary is size %s. "
                "Need logits of shape: (time, vocabulary)" % (tuple(batch.shape[1:]), n_labels)
            )
        if getattr(batch, "is_cuda", False):
            import torch

            native = {torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.bfloat16: 3}
            if batch.dtype not in native:
                return False
            t = batch if batch.is_contiguous() else batch.contiguous()
            # (after the possible layout copy, on the batch's own device: see the generic path)
            torch.cuda.current_stream(batch.device).synchronize()
            self.device_index = batch.device.index
            base, step = t.data_ptr(), t.stride(0) * t.element_size()
            self.dtype = native[batch.dtype]
            self.is_device = True
        elif isinstance(batch, np.ndarray):
            target, self.dtype = {np.dtype(np.float32): (np.float32, 0), np.dtype(np.float16): (np.float16, 2)}.get(
                batch.dtype, (np.float64, 1))
            t = np.ascontiguousarray(batch, dtype=target)
            base, step = t.ctypes.data, t.strides[0]
        else:
            return False
        n, frames = int(t.shape[0]), int(t.shape[1])
        self.keep.append(t)
        # (numpy, not 2 x B Python ints: at 4096 utterances the lists and their ctypes copies cost ~1 ms per call)
        self.ptrs = (np.uint64(base) + np.arange(n, dtype=np.uint64) * np.uint64(step)) if n else []
        self.frames = np.full(n, frames, dtype=np.int32) if n else []
        return True


def _c_arrays(batch: _Batch):
    """(void*[n], int32[n]) for the C ABI from a batch's pointer / frame lists (numpy arrays are passed as they are)."""
    n = len(batch.ptrs)
    if isinstance(batch.ptrs, np.ndarray):
        batch.keep.append((batch.ptrs, batch.frames))
        return (batch.ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), batch.frames.ctypes.data_as(C.POINTER(C.c_int32)))
    return (C.c_void_p * max(n, 1))(*batch.ptrs), (C.c_int32 * max(n, 1))(*batch.frames)


class _DeviceStreams:
    """n streams advancing together on the device (one ctcdec_stream handle) + what the Python side has to remember to
    turn their beams into LMBeams on demand."""

    def __init__(self, decoder: "BeamSearchDecoderCTC", n: int):
        self.decoder = decoder
        self.lib = decoder._lib
        self.n = n
        self.gen = 0
        self.hot_key: Optional[Tuple[str, ...]] = None
        self.params: Optional[B.Params] = None
        self.memos: List[Dict[Any, Any]] = []
        self.parents: Optional[List[List[Beam]]] = None  # the caller's beams of the last import
        self.lists: List[Any] = []  # weak references to the lazy lists of the current generation
        h = C.c_void_p()
        self.lib.check(self.lib.dll.ctcdec_stream_open(decoder._handle, n, None, C.byref(h)))
        self.handle = h

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.lib.dll.ctcdec_stream_close(h)
            except Exception:  # pragma: no cover
                pass
            self.handle = None

    def alive(self, decoder) -> bool:
        return self.handle is not None and self.decoder is decoder

    # -- the slow way in: beams the caller built or edited (decoder.py:681-728 takes any List[Beam]) -------------
    def import_beams(self, beams_list, cached_lm_scores_list) -> None:
        dec = self.decoder
        n_lms = len(dec._members)
        has_lm = n_lms > 0
        vocab2idx = dec._vocab2idx
        pieces: List[bytes] = []
        pos = 0
        total = sum(len(b) for b in beams_list)
        arr = (B.BeamIn * max(total, 1))()
        more = (B.LmState * max(total * (n_lms - 1), 1))() if n_lms > 1 else None  # model 1..'s states per beam
        beam_off = np.zeros(self.n + 1, dtype=np.int64)
        k = 0
        parents = []
        for u in range(self.n):
            beams = list(beams_list[u])
            parents.append(beams)
            for beam in beams:
                text = beam.text if not beam.next_word else (beam.text + " " + beam.next_word if beam.text else beam.next_word)
                text = " ".join(text.split())
                e = arr[k]
                e.logit_score = float(beam.logit_score)
                if has_lm:
                    raw, state = dec._memo_entry(cached_lm_scores_list[u], text)
                    if n_lms > 1:
                        if not isinstance(state, MultiLanguageModelState) or len(state.states) != n_lms:
                            raise AssertionError(
                                f"Wrong input state type found. Expected MultiLanguageModelState of {n_lms}, got {type(state)}")
                        parts = state.states
                        for j in range(1, n_lms):
                            more[k * (n_lms - 1) + j - 1] = parts[j].state.to_c()
                        e.more_states = C.cast(C.byref(more, k * (n_lms - 1) * C.sizeof(B.LmState)), C.POINTER(B.LmState))
                        state = parts[0]
                    if not isinstance(state, KenlmState):
                        raise AssertionError(f"Wrong input state type found. Expected KenlmState, got {type(state)}")
                    e.raw_lm_score = float(raw)
                    e.lm_state = state.state.to_c()
                if beam.last_char is None:
                    e.last_char = -1
                else:
                    if beam.last_char not in vocab2idx:
                        raise ValueError("beam.last_char %r is not a label of this decoder" % (beam.last_char,))
                    e.last_char = vocab2idx[beam.last_char]
                e.partial_start = int(beam.partial_frames[0])
                e.partial_end_frame = int(beam.partial_frames[1])
                tb = text.encode("utf-8")
                pb = beam.partial_word.encode("utf-8")
                e.text_begin, e.text_end = pos, pos + len(tb)
                pos += len(tb)
                e.partial_begin, e.partial_end = pos, pos + len(pb)
                pos += len(pb)
                pieces.append(tb)
                pieces.append(pb)
                k += 1
            beam_off[u + 1] = k
        blob = b"".join(pieces) or b"\0"
        self.lib.check(self.lib.dll.ctcdec_stream_import(self.handle, arr, B.off_ptr(beam_off), blob, pos))
        self.parents = parents

    # -- results ---------------------------------------------------------------------------------------------------
    def retire_lists(self) -> None:
        """The device is about to move on: lists of the current generation that nobody has looked at become unreadable."""
        self.gen += 1
        self.lists = []

    def lazy_lists(self) -> List[List[LMBeam]]:
        import weakref

        out = [_ResidentBeams(self, u, self.gen) for u in range(self.n)]
        self.lists = [weakref.ref(x) for x in out]
        return out

    def fill_current(self) -> None:
        """One native read materialises the beams of every stream (the lazy lists of this generation that are still alive)."""
        dec = self.decoder
        with dec._call_lock:
            res = C.c_void_p()
            self.lib.check(self.lib.dll.ctcdec_stream_read(self.handle, C.byref(self.params), C.byref(res)))
            try:
                got = self.unpack(res)
            finally:
                self.lib.dll.ctcdec_result_free(res)
        for u, ref in enumerate(self.lists):
            lst = ref()
            if lst is not None and not lst._filled:
                list.extend(lst, got[u])
                lst._filled = True

    def peek_best(self) -> None:
        """`beams[0]` on an unread list (a streaming caller showing the transcript so far): one native read with
        n_best = 1 fetches the best beam of every stream; the lists stay unread otherwise."""
        dec = self.decoder
        with dec._call_lock:
            p = B.Params.from_buffer_copy(self.params)
            p.n_best = 1
            res = C.c_void_p()
            self.lib.check(self.lib.dll.ctcdec_stream_read(self.handle, C.byref(p), C.byref(res)))
            try:
                got = self.unpack(res)
            finally:
                self.lib.dll.ctcdec_result_free(res)
        for u, ref in enumerate(self.lists):
            lst = ref()
            if lst is not None and not lst._filled:
                lst._best = got[u][0] if got[u] else None
                lst._peeked = True

    def unpack(self, res) -> List[List[LMBeam]]:
        dec = self.decoder
        lib = self.lib
        n = self.n
        n_lms = len(dec._members)
        has_lm = n_lms > 0
        pk = B.Packed()
        lib.check(lib.dll.ctcdec_result_pack(res, C.byref(pk)))
        nb, nw = int(pk.n_beams), int(pk.n_words)
        if nb == 0:
            return [[] for _ in range(n)]
        b_off = np.ctypeslib.as_array(pk.beam_off, shape=(n + 1,))
        if self.parents is None and not os.environ.get("CTCDEC_PY_UNPACK"):
            # the same lists from one C loop (csrc/pytexts.c); the memo entries of new texts are added here
            frames_of = None
            if not os.environ.get("CTCDEC_EAGER_FRAMES"):  # (diagnostics / tests: plain lists of tuples built in C)
                if nw:
                    ws = np.ctypeslib.as_array(pk.word_start, shape=(nw,)).copy()
                    we = np.ctypeslib.as_array(pk.word_end, shape=(nw,)).copy()
                else:
                    ws = we = np.zeros(0, dtype=np.int32)
                frames_of = _lazy_frames_factory(ws, we)
            built = B.lm_beams(LMBeam, n, pk, dec._labels_list, frames_of)
            if built is not None:
                if has_lm:
                    self._note_memo_entries(res, pk, built, b_off, nb, n_lms)
                return built
        t_off = np.ctypeslib.as_array(pk.text_off, shape=(nb + 1,))
        tblob = C.string_at(pk.text_blob, int(t_off[nb])) if t_off[nb] else b""
        p_off = np.ctypeslib.as_array(pk.partial_off, shape=(nb + 1,))
        pblob = C.string_at(pk.partial_blob, int(p_off[nb])) if p_off[nb] else b""
        logit = np.ctypeslib.as_array(pk.logit_score, shape=(nb,))
        lms = np.ctypeslib.as_array(pk.lm_score, shape=(nb,))
        raw = np.ctypeslib.as_array(pk.raw_lm_score, shape=(nb,))
        src = np.ctypeslib.as_array(pk.src_beam, shape=(nb,))
        lch = np.ctypeslib.as_array(pk.last_char, shape=(nb,))
        ps = np.ctypeslib.as_array(pk.partial_start, shape=(nb,))
        pe = np.ctypeslib.as_array(pk.partial_end, shape=(nb,))
        wco = np.ctypeslib.as_array(pk.word_cnt_off, shape=(nb + 1,))
        if nw:
            wstart = np.ctypeslib.as_array(pk.word_start, shape=(nw,))
            wend = np.ctypeslib.as_array(pk.word_end, shape=(nw,))
        out: List[List[LMBeam]] = []
        for u in range(n):
            outs = []
            memo = self.memos[u] if u < len(self.memos) else {}
            for j in range(int(b_off[u]), int(b_off[u + 1])):
                text = tblob[int(t_off[j]) : int(t_off[j + 1])].decode("utf-8")
                partial = pblob[int(p_off[j]) : int(p_off[j + 1])].decode("utf-8")
                new_frames = [(int(wstart[w]), int(wend[w])) for w in range(int(wco[j]), int(wco[j + 1]))]
                if src[j] >= 0 and self.parents is not None:  # decoded below one of the caller's own beams
                    new_frames = list(self.parents[u][int(src[j])].text_frames) + new_frames
                last = None if lch[j] < 0 else dec._idx2vocab[int(lch[j])]
                outs.append(LMBeam(text, "", partial, last, new_frames, (int(ps[j]), int(pe[j])), float(logit[j]), float(lms[j])))
                if has_lm and (text, False) not in memo:
                    memo[(text, False)] = (float(raw[j]), float(raw[j]), self._state_of(res, pk, u, j, j - int(b_off[u]), n_lms))
            out.append(outs)
        return out

    def _note_memo_entries(self, res, pk, built, b_off, nb: int, n_lms: int) -> None:
        """The caller's caches get an entry for every returned text, as the reference's decode leaves them (decoder.py:387-396).
        A cache that get_starting_state() handed out (_LazyMemo) takes ONE note per read and makes the entries when somebody
        looks; a caller's own dict is filled here. The entries' LM-state objects are made on demand from a private copy of the
        read's packed states (several LMs: eagerly, one native call per beam -- rare, short lists)."""
        raws = np.ctypeslib.as_array(pk.raw_lm_score, shape=(nb,)).tolist()
        sz = C.sizeof(B.LmState)
        whole = C.string_at(C.cast(pk.lm_state, C.c_void_p), nb * sz)
        for u in range(self.n):
            memo = self.memos[u] if u < len(self.memos) else {}
            j0 = int(b_off[u])
            beams = built[u]
            # (a stream's entries keep only that stream's slice of the read's packed states and scores alive)
            store = _StateStore(whole[j0 * sz:(j0 + len(beams)) * sz])
            if n_lms == 1 and type(memo) is _LazyMemo:
                memo._note(beams, raws[j0:j0 + len(beams)], store)
                continue
            j = j0
            for beam in beams:
                key = (beam.text, False)
                if key not in memo:
                    if n_lms > 1:
                        memo[key] = (raws[j], raws[j], self._state_of(res, pk, u, j, j - j0, n_lms))
                    else:
                        memo[key] = _MemoEntry.make(raws[j], store, j - j0)
                j += 1

    def _state_of(self, res, pk, u: int, j: int, j_in_stream: int, n_lms: int) -> AbstractLMState:
        state: AbstractLMState = KenlmState(NgramState.from_c(pk.lm_state[j]))
        if n_lms > 1:
            parts = [state]
            for x in range(1, n_lms):
                cst = B.LmState()
                self.lib.check(self.lib.dll.ctcdec_result_lm_state_of(res, u, j_in_stream, x, C.byref(cst)))
                parts.append(KenlmState(NgramState.from_c(cst)))
            state = MultiLanguageModelState(parts)
        return state


class _LazyFrames(list):
    """text_frames of a returned beam -- an LMBeam's (start, end) pairs, an OutputBeam's (word, (start, end)) pairs --, kept as a
    window of the result's two
    int32 arrays until somebody looks (then an ordinary list of tuples, equal to what the reference returns,
    decoder.py:653-667). A stream that has run for a thousand frames carries ~250 words per beam; building their tuples for
    every beam of every stream was most of the time of reading the beams (round 4: 7.8 ms for 64 streams, 11.3 ms with the
    last chunk), and a caller that hands the beams back, shows the best text or looks at one beam's frames never needs them.
    Same caveat as _ResidentBeams for code that reads list storage through the C API without calling a method."""

    __slots__ = ("_src", "_lo", "_hi", "_text")

    def __init__(self, *a) -> None:
        # (an instance made as type(frames)(iterable) -- dataclasses.asdict and copy-like idioms rebuild list fields that way --
        #  is an ordinary, already filled list)
        self._src = None
        self._lo = self._hi = 0
        self._text = None
        list.__init__(self, *a)

    def _fill(self) -> None:
        src = self._src
        if src is not None:
            self._src = None
            ws, we = src
            spans = zip(ws[self._lo:self._hi].tolist(), we[self._lo:self._hi].tolist())
            text = self._text
            if text is None:  # an LMBeam's frames: (start, end) per word
                list.extend(self, spans)
            else:  # an OutputBeam's: (word, (start, end)) (decoder.py:653-667)
                self._text = None
                list.extend(self, zip(text.split(" ") if text else (), spans))

    _touch = _fill

    def __reduce__(self):
        self._fill()
        return (list, (list(list.__iter__(self)),))


def _lazy_frames_factory(word_start, word_end):
    """frames_of(w0, w1) over private copies of a result's word_start / word_end arrays (the result is freed after the read)"""
    src = (word_start, word_end)

    def frames_of(lo, hi):
        f = _LazyFrames()
        f._src = src
        f._lo = lo
        f._hi = hi
        f._text = None
        return f

    return frames_of


def _lazy_word_frames_factory(word_start, word_end):
    """frames_of(text, w0, w1) for OutputBeams: the words of `text` paired with their frames when somebody looks"""
    src = (word_start, word_end)

    def frames_of(text, lo, hi):
        f = _LazyFrames()
        f._src = src
        f._lo = lo
        f._hi = hi
        f._text = text
        return f

    return frames_of


class _StateStore:
    """The packed LM states of one device read (a private copy: the result is freed after the read) and the state objects made
    from them so far."""

    __slots__ = ("_bytes", "_made")

    def __init__(self, state_bytes: bytes):
        self._bytes = state_bytes
        self._made: Dict[int, AbstractLMState] = {}

    def state(self, j: int) -> AbstractLMState:
        st = self._made.get(j)
        if st is None:
            st = KenlmState(NgramState.from_c(B.LmState.from_buffer_copy(self._bytes, j * C.sizeof(B.LmState))))
            self._made[j] = st
        return st


class _MemoEntry(tuple):
    """A memo entry (lm_score + hot-word score, raw lm_score, LM state) of a text a device read returned -- the tuple the
    reference's cache holds (decoder.py:387-396) -- whose STATE object is built from the read's packed states when somebody
    unpacks or indexes the entry (the import path of edited beams does; a caller that only hands its caches back never does).
    Storage: (raw, raw, store, index); every way of looking sees the three-tuple."""

    __slots__ = ()

    @staticmethod
    def make(raw, store: _StateStore, j: int) -> "_MemoEntry":
        return tuple.__new__(_MemoEntry, (raw, raw, store, j))

    def _full(self):
        g = tuple.__getitem__
        return (g(self, 0), g(self, 1), g(self, 2).state(g(self, 3)))

    def __len__(self):
        return 3

    def __getitem__(self, k):
        return self._full()[k]

    def __iter__(self):
        return iter(self._full())

    def __eq__(self, other):
        return self._full() == (other._full() if isinstance(other, _MemoEntry) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None  # type: ignore[assignment]

    def __repr__(self):
        return repr(self._full())

    def __reduce__(self):
        return (tuple, (self._full(),))


class _LazyMemo(dict):
    """`cached_lm_scores` as get_starting_state() hands it out (decoder.py:669-679): a dict that remembers what the device reads
    of its stream returned -- one O(1) note per read -- and turns the notes into entries when somebody LOOKS (any dict method;
    the import path of edited beams does). The reference fills this cache as a side effect of decoding; doing that eagerly was
    a tuple, a dict probe and an entry object per returned beam on every read, for a caller that almost never looks.
    A caller's own plain dict is filled eagerly as before."""

    __slots__ = ("_pending",)

    def __init__(self, *a, **k):
        dict.__init__(self, *a, **k)
        self._pending: List[Any] = []

    _MAX_PENDING = 8

    def _note(self, beams, raws, store: _StateStore) -> None:
        # a snapshot of the read's beams (the caller may sort / delete in the list it was handed; entries pair beams with the
        # stream's slice of the read's raw scores and states BY POSITION); a stream that is read
        # every chunk and never looks at its cache settles after a few notes instead of pinning every read it ever made
        self._pending.append((tuple(beams), raws, store))
        if len(self._pending) > self._MAX_PENDING:
            self._settle()

    def _settle(self) -> None:
        pend = self._pending
        if pend:
            self._pending = []
            has, put, make = dict.__contains__, dict.__setitem__, _MemoEntry.make
            for beams, raws, store in pend:
                for k, beam in enumerate(beams):
                    key = (beam.text, False)
                    if not has(self, key):
                        put(self, key, make(raws[k], store, k))

    def __reduce__(self):
        self._settle()
        return (dict, (dict(dict.items(self)),))


def _settling(name):
    def method(self, *a, **k):
        self._settle()
        return getattr(dict, name)(self, *a, **k)

    method.__name__ = name
    return method


for _n in ("__getitem__", "__contains__", "__iter__", "__len__", "__repr__", "__eq__", "__ne__", "get", "keys", "values", "items",
           "copy", "pop", "popitem", "setdefault", "update", "__delitem__", "clear", "__or__", "__ror__", "__ior__", "__reversed__"):
    setattr(_LazyMemo, _n, _settling(_n))
_LazyMemo.__hash__ = None  # type: ignore[assignment]


class _ResidentBeams(list):
    """The LMBeams of one stream after a chunk, still on the device: an ordinary list that fills itself the first time it
    is looked at. Handed back unchanged to partial_decode_beams(_batch) it is never filled at all.
    Caveat: code that reads a list through the C API without calling a method (PySequence_Fast / PyList_GET_SIZE on a
    subclass: the C json encoder, some extension modules) sees the unfilled storage -- pass list(beams) to such code.
    Pickling and copying go through __reduce__ and yield a plain, filled list."""

    def __init__(self, streams: _DeviceStreams, index: int, gen: int):
        super().__init__()
        self._streams = streams
        self._index = index
        self._gen = gen
        self._filled = False
        self._edited = False
        self._peeked = False
        self._best: Optional[LMBeam] = None

    def __getitem__(self, key):
        # the best beam alone is a cheap read (n_best = 1); anything else fills the list
        if not self._filled and type(key) is int and key == 0 and self._gen == self._streams.gen:
            if not self._peeked:
                self._streams.peek_best()
            if self._best is not None:
                return self._best
        self._fill()
        return list.__getitem__(self, key)

    def _current(self) -> bool:
        return not self._edited and self._gen == self._streams.gen

    def __reduce__(self):
        # pickle / copy / multiprocessing: as the plain list of its beams (the device handle stays behind)
        self._fill()
        return (list, (list(list.__iter__(self)),))

    def _fill(self) -> None:
        if self._filled:
            return
        if self._gen != self._streams.gen:
            raise RuntimeError(
                "these beams were handed back to partial_decode_beams and the stream has moved on: read them before the "
                "next call, or keep list(beams) (CTCDEC_RESIDENT_STREAMS=0 returns plain lists every time)")
        self._streams.fill_current()

    def _touch(self) -> None:
        self._fill()
        self._edited = True


def _fill_args(args) -> None:
    # (the list type's own C code reads another lazy list's storage directly: a comparison / concatenation / extend with one
    #  has to fill it first)
    for x in args:
        if isinstance(x, (_LazyFrames, _ResidentBeams)):
            x._fill()


def _reader(name):
    def method(self, *a, **k):
        self._fill()
        _fill_args(a)
        return getattr(list, name)(self, *a, **k)

    method.__name__ = name
    return method


def _writer(name):
    def method(self, *a, **k):
        self._touch()
        _fill_args(a)
        return getattr(list, name)(self, *a, **k)

    method.__name__ = name
    return method


for _n in ("__len__", "__iter__", "__contains__", "__repr__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__",
           "__ge__", "__add__", "__mul__", "__rmul__", "__reversed__", "index", "count", "copy"):
    setattr(_ResidentBeams, _n, _reader(_n))
for _n in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove", "clear", "sort",
           "reverse"):
    setattr(_ResidentBeams, _n, _writer(_n))
_ResidentBeams.__hash__ = None  # type: ignore[assignment]


def _radd(self, other):
    # plain_list + lazy: a subclass's reflected method is tried first, which is the only chance to fill before list's C code
    # concatenates the (still empty) storage
    self._fill()
    return list(other) + list(list.__iter__(self))


_ResidentBeams.__radd__ = _radd  # type: ignore[attr-defined]
_LazyFrames.__radd__ = _radd  # type: ignore[attr-defined]
for _n in ("__len__", "__iter__", "__contains__", "__repr__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__add__",
           "__mul__", "__rmul__", "__reversed__", "index", "count", "copy", "__getitem__", "__setitem__", "__delitem__", "__iadd__",
           "__imul__", "append", "extend", "insert", "pop", "remove", "clear", "sort", "reverse"):
    setattr(_LazyFrames, _n, _reader(_n))
_LazyFrames.__hash__ = None  # type: ignore[assignment]


class BeamSearchDecoderCTC:
    # The reference parks the language model in a class-level dict keyed by 16 random bytes so that
    # fork-pool children find it (decoder.py:262-269); the container and its clean-up functions
    # are kept because callers and tests manage memory through them.
    model_container: Dict[bytes, Optional[AbstractLanguageModel]] = {}

    def __init__(self, alphabet: Alphabet, language_model: Optional[AbstractLanguageModel] = None) -> None:
        self._alphabet = alphabet
        self._idx2vocab = {n: c for n, c in enumerate(self._alphabet.labels)}
        self._labels_list = list(self._alphabet.labels)
        self._vocab2idx = {c: n for n, c in enumerate(self._alphabet.labels)}
        self._is_bpe = alphabet.is_bpe
        self._model_key = os.urandom(16)
        BeamSearchDecoderCTC.model_container[self._model_key] = language_model
        members: List[LanguageModel] = []
        if isinstance(language_model, MultiLanguageModel):
            members = language_model.language_models
            if len(members) > MAX_LANGUAGE_MODELS:
                raise NotImplementedError("a MultiLanguageModel of more than %d models" % MAX_LANGUAGE_MODELS)
        elif language_model is not None:
            members = [language_model]
        if not all(isinstance(m, LanguageModel) for m in members):
            raise NotImplementedError(
                "only n-gram LanguageModels (alone or inside a MultiLanguageModel) can be lowered to the "
                "device trie; user-defined AbstractLanguageModel subclasses are not supported "
                "(there is deliberately no CPU fallback)"
            )
        self._members = members
        self._device = _default_device()  # one process per GPU: LOCAL_RANK, or CTCDEC_DEVICE when set
        lib = B.get_library()
        self._lib = lib
        blob, off = B.pack_strings(self._alphabet.labels)
        handle = C.c_void_p()
        lib.check(
            lib.dll.ctcdec_create(blob, B.off_ptr(off), len(self._alphabet.labels), int(self._is_bpe),
                                  self._device, C.byref(handle))
        )
        self._handle = handle
        # the device the library actually bound (one visible GPU per rank under SLURM / HIP_VISIBLE_DEVICES: LOCAL_RANK may
        # exceed the device count and the library falls back to device 0)
        self._device = int(lib.dll.ctcdec_device())
        if len(members) == 1:
            lib.check(lib.dll.ctcdec_lm_share(handle, members[0]._kenlm_model._handle))
        elif len(members) > 1:
            srcs = (C.c_void_p * len(members))(*[m._kenlm_model._handle for m in members])
            lib.check(lib.dll.ctcdec_lm_share_multi(handle, srcs, len(members)))
        self._hot_key: Optional[Tuple[str, ...]] = None
        # a byte no decoded text can contain (texts are made of label characters and spaces): lets decode_batch take its
        # texts as one joined buffer and split it natively
        self._texts_sep: Optional[bytes] = next(
            (c.encode("ascii") for c in ("\n", "\x01", "\x02") if not any(c in lab for lab in self._alphabet.labels)), None)
        # one native call at a time per decoder: hot words / per-model weights are decoder state that a call sets
        # up first (host threads may share a decoder; ctypes releases the GIL while the device works)
        self._call_lock = threading.RLock()

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                self._lib.dll.ctcdec_destroy(h)
            except Exception:  # pragma: no cover
                pass
            self._handle = None

    # -- parameter / model management (decoder.py:292-328) -------------------------------------
    def reset_params(
        self,
        alpha: Optional[float] = None,
        beta: Optional[float] = None,
        unk_score_offset: Optional[float] = None,
        lm_score_boundary: Optional[bool] = None,
    ) -> None:
        """Forward the weights that were given to the language model (decoder.py:292-316); no model, nothing to do."""
        lm = self._language_model
        if lm is None:
            return
        given = {"alpha": alpha, "beta": beta, "unk_score_offset": unk_score_offset, "score_boundary": lm_score_boundary}
        lm.reset_params(**{name: value for name, value in given.items() if value is not None})

    @classmethod
    def clear_class_models(cls) -> None:
        cls.model_container = {}

    def cleanup(self) -> None:
        if self._model_key in BeamSearchDecoderCTC.model_container:
            del BeamSearchDecoderCTC.model_container[self._model_key]

    @property
    def _language_model(self) -> Optional[AbstractLanguageModel]:
        return BeamSearchDecoderCTC.model_container[self._model_key]

    def _check_logits_dimension(self, logits: Any) -> None:
        if len(logits.shape) != 2:
            raise ValueError(
                "Input logits have %s dimensions, but need 2: (time, vocabulary)" % len(logits.shape)
            )
        if logits.shape[-1] != len(self._idx2vocab):
            raise ValueError(
                "Input logits shape is %s, but vocabulary is size %s. "
                "Need logits of shape: (time, vocabulary)" % (tuple(logits.shape), len(self._idx2vocab))
            )

    # -- the one native call everything funnels into --------------------------------------------
    def _set_hotwords(self, hotwords: Optional[Iterable[str]]) -> None:
        unigrams = HotwordScorer.build_scorer(hotwords).unigrams
        key = tuple(unigrams)
        if key == self._hot_key:
            return
        blob, off = B.pack_strings(unigrams)
        self._lib.check(self._lib.dll.ctcdec_set_hotwords(self._handle, blob, B.off_ptr(off), len(unigrams)))
        self._hot_key = key

    def _params(self, beam_width, beam_prune_logp, token_min_logp, prune_history, hotword_weight, n_best) -> B.Params:
        lm = self._members[0] if self._members else None  # model 0; the others go through ctcdec_lm_set_params
        with self._call_lock:  # (RLock: the streaming path already holds it; nothing may change the handle mid-call)
            for k, m in enumerate(self._members[1:], start=1):
                self._lib.check(self._lib.dll.ctcdec_lm_set_params(
                    self._handle, k, float(m.alpha), float(m.beta), float(m.unk_score_offset), int(bool(m.score_boundary))))
        p = B.Params()
        p.beam_width = int(beam_width)
        p.prune_history = int(bool(prune_history))
        p.n_best = int(n_best)
        p.want_lm_state = 1
        p.beam_prune_logp = float(beam_prune_logp)
        p.token_min_logp = float(token_min_logp)
        p.hotword_weight = float(hotword_weight)
        p.alpha = float(lm.alpha) if lm is not None else 0.0
        p.beta = float(lm.beta) if lm is not None else 0.0
        p.unk_score_offset = float(lm.unk_score_offset) if lm is not None else 0.0
        p.log_base_change = LOG_BASE_CHANGE_FACTOR
        p.lm_score_boundary = int(bool(lm.score_boundary)) if lm is not None else 0
        p.first_frame = 0
        p.texts_only = 0
        p.reserved = 0
        return p

    def _run(self, logits_list: Sequence[Any], params: B.Params, hotwords, start_states=None):
        """-> result handle. Caller must free the handle."""
        with self._call_lock:
            return self._run_locked(logits_list, params, hotwords, start_states)

    def _run_locked(self, logits_list: Sequence[Any], params: B.Params, hotwords, start_states=None):
        self._set_hotwords(hotwords)
        batch = _Batch(logits_list, len(self._idx2vocab))
        if batch.is_device and batch.device_index is not None and batch.device_index != self._device:
            raise ValueError("the logits live on cuda:%d but this decoder was built for cuda:%d (one process per GPU: "
                             "LOCAL_RANK / CTCDEC_DEVICE pick the device)" % (batch.device_index, self._device))
        n = len(batch.ptrs)
        ptrs, frames = _c_arrays(batch)
        st_arr = None
        n_lms = len(self._members)
        if start_states is not None and n_lms > 0:
            st_arr = (B.LmState * max(n * n_lms, 1))()
            for k, s in enumerate(start_states):
                if s is None:
                    for j in range(n_lms):
                        st_arr[k * n_lms + j].length = -1
                    continue
                if n_lms > 1:  # language_model.py:488-497
                    if not isinstance(s, MultiLanguageModelState):
                        raise AssertionError(
                            f"Wrong input state type found. Expected MultiLanguageModelState, got {type(s)}")
                    if len(s.states) != n_lms:
                        raise AssertionError(
                            f"Number of states ({len(s.states)}) does not match number of language models ({n_lms}).")
                    parts = s.states
                else:
                    parts = [s]
                for j, part in enumerate(parts):
                    if not isinstance(part, KenlmState):
                        raise AssertionError(f"Wrong input state type found. Expected KenlmState, got {type(part)}")
                    st_arr[k * n_lms + j] = part.state.to_c()
        res = C.c_void_p()
        self._lib.check(
            self._lib.dll.ctcdec_decode_batch(self._handle, ptrs, frames, n, batch.dtype, int(batch.is_device),
                                              C.byref(params), st_arr, C.byref(res))
        )
        ms = (C.c_double * 3)()
        self._lib.dll.ctcdec_result_timing(res, ms)
        # [frame-prune kernels, beam kernel (HIP events on the decode stream), whole native call]
        self.last_timing_ms = (float(ms[0]), float(ms[1]), float(ms[2]))
        # 1: one wavefront per utterance (beam_wave.h), 2: one workgroup per utterance (beam_core.h)
        self.last_beam_kernel = int(self._lib.dll.ctcdec_result_beam_kernel(res))
        return res

    def _unpack(self, res: C.c_void_p, with_state: bool) -> List[List[OutputBeam]]:
        pk = B.Packed()
        self._lib.check(self._lib.dll.ctcdec_result_pack(res, C.byref(pk)))
        nb, nw, nu = int(pk.n_beams), int(pk.n_words), int(pk.n_utts)
        beam_off = np.ctypeslib.as_array(pk.beam_off, shape=(nu + 1,))
        out: List[List[OutputBeam]] = []
        if nb == 0:
            return [[] for _ in range(nu)]
        has_lm = self._language_model is not None
        if not os.environ.get("CTCDEC_PY_UNPACK"):  # the same lists from one C loop (csrc/pytexts.c)
            states = None
            if with_state and has_lm:
                states = []
                for u in range(nu):
                    for k in range(int(beam_off[u]), int(beam_off[u + 1])):
                        states.append(self._beam_state(res, pk, u, k, k - int(beam_off[u])))
            frames_of = None
            if not os.environ.get("CTCDEC_EAGER_FRAMES"):  # (diagnostics / tests: plain lists of tuples built in C)
                if nw:
                    ws = np.ctypeslib.as_array(pk.word_start, shape=(nw,)).copy()
                    we = np.ctypeslib.as_array(pk.word_end, shape=(nw,)).copy()
                else:
                    ws = we = np.zeros(0, dtype=np.int32)
                frames_of = _lazy_word_frames_factory(ws, we)
            built = B.output_beams(OutputBeam, pk, states, frames_of)
            if built is not None:
                return built
        text_off = np.ctypeslib.as_array(pk.text_off, shape=(nb + 1,))
        blob = C.string_at(pk.text_blob, int(text_off[nb])) if text_off[nb] else b""
        logit = np.ctypeslib.as_array(pk.logit_score, shape=(nb,))
        lm = np.ctypeslib.as_array(pk.lm_score, shape=(nb,))
        wco = np.ctypeslib.as_array(pk.word_cnt_off, shape=(nb + 1,))
        if nw:
            wstart = np.ctypeslib.as_array(pk.word_start, shape=(nw,))
            wend = np.ctypeslib.as_array(pk.word_end, shape=(nw,))
        for u in range(nu):
            beams = []
            for k in range(int(beam_off[u]), int(beam_off[u + 1])):
                text = blob[int(text_off[k]) : int(text_off[k + 1])].decode("utf-8")
                words = text.split(" ") if text else []
                w0, w1 = int(wco[k]), int(wco[k + 1])
                frames = [(words[j], (int(wstart[w0 + j]), int(wend[w0 + j]))) for j in range(w1 - w0)]
                state = self._beam_state(res, pk, u, k, k - int(beam_off[u])) if (with_state and has_lm) else None
                beams.append(OutputBeam(text, state, frames, float(logit[k]), float(lm[k])))
            out.append(beams)
        return out

    def _beam_state(self, res, pk, u: int, k: int, j_in_utt: int) -> AbstractLMState:
        """last_lm_state of packed beam k (beam j_in_utt of utterance u)."""
        state: AbstractLMState = KenlmState(NgramState.from_c(pk.lm_state[k]))
        if len(self._members) > 1:
            parts = [state]
            for j in range(1, len(self._members)):
                cst = B.LmState()
                self._lib.check(self._lib.dll.ctcdec_result_lm_state_of(res, u, j_in_utt, j, C.byref(cst)))
                parts.append(KenlmState(NgramState.from_c(cst)))
            state = MultiLanguageModelState(parts)
        return state

    # -- public decode surface (decoder.py:730-945) ----------------------------------------------
    def decode_beams(
        self,
        logits: Any,
        beam_width: int = DEFAULT_BEAM_WIDTH,
        beam_prune_logp: float = DEFAULT_PRUNE_LOGP,
        token_min_logp: float = DEFAULT_MIN_TOKEN_LOGP,
        prune_history: bool = DEFAULT_PRUNE_BEAMS,
        hotwords: Optional[Iterable[str]] = None,
        hotword_weight: float = DEFAULT_HOTWORD_WEIGHT,
        lm_start_state: Optional[AbstractLMState] = None,
    ) -> List[OutputBeam]:
        self._check_logits_dimension(logits)
        params = self._params(beam_width, beam_prune_logp, token_min_logp, prune_history, hotword_weight, 0)
        res = self._run([logits], params, hotwords, [lm_start_state])
        try:
            return self._unpack(res, True)[0]
        finally:
            self._lib.dll.ctcdec_result_free(res)

    def decode(
        self,
        logits: Any,
        beam_width: int = DEFAULT_BEAM_WIDTH,
        beam_prune_logp: float = DEFAULT_PRUNE_LOGP,
        token_min_logp: float = DEFAULT_MIN_TOKEN_LOGP,
        hotwords: Optional[Iterable[str]] = None,
        hotword_weight: float = DEFAULT_HOTWORD_WEIGHT,
        lm_start_state: Optional[AbstractLMState] = None,
    ) -> str:
        self._check_logits_dimension(logits)
        # prune_history=True: only the best beam is read (decoder.py:888)
        params = self._params(beam_width, beam_prune_logp, token_min_logp, True, hotword_weight, 1)
        res = self._run([logits], params, hotwords, [lm_start_state])
        try:
            return self._unpack(res, False)[0][0].text
        finally:
            self._lib.dll.ctcdec_result_free(res)

    def decode_batch(
        self,
        pool: Any,
        logits_list: Sequence[Any],
        beam_width: int = DEFAULT_BEAM_WIDTH,
        beam_prune_logp: float = DEFAULT_PRUNE_LOGP,
        token_min_logp: float = DEFAULT_MIN_TOKEN_LOGP,
        hotwords: Optional[Iterable[str]] = None,
        hotword_weight: float = DEFAULT_HOTWORD_WEIGHT,
    ) -> List[str]:
        """decoder.py:895-945. One device launch decodes the whole batch: a multiprocessing pool is ignored. A
        `pyctcdecode_amd.parallel.DevicePool` -- one worker process per GPU -- shards the batch over its devices."""
        if getattr(logits_list, "ndim", 0) != 3:
            logits_list = list(logits_list)
        if len(logits_list) == 0:
            return []
        if type(pool).__name__ == "DevicePool":
            return pool.decode_batch(logits_list, beam_width=beam_width, beam_prune_logp=beam_prune_logp,
                                     token_min_logp=token_min_logp, hotwords=hotwords, hotword_weight=hotword_weight)
        params = self._params(beam_width, beam_prune_logp, token_min_logp, True, hotword_weight, 1)
        params.texts_only = 1  # (the kernels write the texts themselves: no emission lists to copy back and replay)
        res = self._run(logits_list, params, hotwords)
        try:
            texts = B.texts_of(self._lib, res)  # (one str per block of the library's memory, built in C)
            if texts is not None:
                return texts
            if self._texts_sep is not None:  # one split instead of one slice per utterance (0.7 -> 0.15 ms at 4096)
                blob_p, nbytes, n = C.c_void_p(), C.c_int64(), C.c_int64()
                self._lib.check(self._lib.dll.ctcdec_result_texts_joined(res, self._texts_sep, C.byref(blob_p), C.byref(nbytes),
                                                                         C.byref(n)))
                if n.value == 0:
                    return []
                parts = B.split_texts(blob_p, int(nbytes.value), int(n.value), self._texts_sep)
                if len(parts) == n.value:
                    return parts
            blob_p, off_p, n = C.c_void_p(), C.POINTER(C.c_int64)(), C.c_int64()
            self._lib.check(self._lib.dll.ctcdec_result_texts(res, C.byref(blob_p), C.byref(off_p), C.byref(n)))
            nb = int(n.value)
            text_off = np.ctypeslib.as_array(off_p, shape=(nb + 1,))
            blob = C.string_at(blob_p, int(text_off[nb])) if text_off[nb] else b""
            off = text_off.tolist()
            if blob.isascii():  # byte offsets == character offsets: decode once, slice the str
                text = blob.decode("ascii")
                return [text[off[k] : off[k + 1]] for k in range(nb)]
            return [blob[off[k] : off[k + 1]].decode("utf-8") for k in range(nb)]
        finally:
            self._lib.dll.ctcdec_result_free(res)

    def decode_beams_batch(
        self,
        pool: Any,
        logits_list: Sequence[Any],
        beam_width: int = DEFAULT_BEAM_WIDTH,
        beam_prune_logp: float = DEFAULT_PRUNE_LOGP,
        token_min_logp: float = DEFAULT_MIN_TOKEN_LOGP,
        prune_history: bool = DEFAULT_PRUNE_BEAMS,
        hotwords: Optional[Iterable[str]] = None,
        hotword_weight: float = DEFAULT_HOTWORD_WEIGHT,
    ) -> List[List[OutputBeam]]:
        """decoder.py:801-857. Beams carry ``last_lm_state=None`` like the reference's mp-safe beams. ``pool``: see decode_batch."""
        logits_list = list(logits_list)
        for logits in logits_list:
            self._check_logits_dimension(logits)
        if len(logits_list) == 0:
            return []
        if type(pool).__name__ == "DevicePool":
            return pool.decode_beams_batch(logits_list, beam_width=beam_width, beam_prune_logp=beam_prune_logp,
                                           token_min_logp=token_min_logp, prune_history=prune_history, hotwords=hotwords,
                                           hotword_weight=hotword_weight)
        params = self._params(beam_width, beam_prune_logp, token_min_logp, prune_history, hotword_weight, 0)
        res = self._run(logits_list, params, hotwords)
        try:
            return self._unpack(res, False)
        finally:
            self._lib.dll.ctcdec_result_free(res)

    # -- serialisation (decoder.py:947-1043): alphabet.json + language_model/ --------------------------
    _ALPHABET_SERIALIZED_FILENAME = "alphabet.json"
    _LANGUAGE_MODEL_SERIALIZED_DIRECTORY = "language_model"

    def save_to_dir(self, filepath: str) -> None:
        """Layout of decoder.py:947-962: <dir>/alphabet.json and, with a language model, <dir>/language_model/."""
        alphabet_file = os.path.join(filepath, self._ALPHABET_SERIALIZED_FILENAME)
        with open(alphabet_file, "w") as out:
            out.write(self._alphabet.dumps())
        if self._language_model is None:
            logger.info("decoder has no language model.")
            return
        lm_dir = os.path.join(filepath, self._LANGUAGE_MODEL_SERIALIZED_DIRECTORY)
        os.makedirs(lm_dir)
        logger.info("Saving language model to %s", lm_dir)
        self._language_model.save_to_dir(lm_dir)

    @staticmethod
    def parse_directory_contents(filepath: str) -> Dict[str, Optional[str]]:
        """Where the parts of a saved decoder are (decoder.py:964-990): {"alphabet": path, "language_model": dir or None}.
        Hidden and dunder entries are ignored; anything else besides the two known names is an error."""
        alphabet_name = BeamSearchDecoderCTC._ALPHABET_SERIALIZED_FILENAME
        lm_name = BeamSearchDecoderCTC._LANGUAGE_MODEL_SERIALIZED_DIRECTORY
        entries = sorted(e for e in os.listdir(filepath) if not e.startswith((".", "__")))
        if alphabet_name not in entries:
            raise ValueError(f"Could not find alphabet file {alphabet_name}. Found {entries}")
        others = [e for e in entries if e != alphabet_name]
        if others and lm_name not in others:
            raise ValueError(f"Could not find language model directory. Looking for {lm_name}, found {others}")
        return {"alphabet": os.path.join(filepath, alphabet_name),
                "language_model": os.path.join(filepath, lm_name) if others else None}

    @classmethod
    def load_from_dir(cls, filepath: str, unigram_encoding: Optional[str] = None) -> "BeamSearchDecoderCTC":
        """Inverse of save_to_dir (decoder.py:992-1005)."""
        parts = cls.parse_directory_contents(filepath)
        with open(parts["alphabet"]) as src:
            alphabet = Alphabet.loads(src.read())
        lm_dir = parts["language_model"]
        lm = None if lm_dir is None else LanguageModel.load_from_dir(lm_dir, unigram_encoding=unigram_encoding)
        return cls(alphabet, language_model=lm)

    @classmethod
    def load_from_hf_hub(cls, model_id: str, cache_dir: Optional[str] = None, **kwargs: Any) -> "BeamSearchDecoderCTC":
        """decoder.py:1007-1043: fetch a saved decoder directory from the Hugging Face hub (default cache
        ~/.cache/pyctcdecode) and load it."""
        try:
            import huggingface_hub
        except ImportError as exc:
            raise ImportError("load_from_hf_hub needs the huggingface_hub package "
                              "(https://pypi.org/project/huggingface-hub/).") from exc
        if cache_dir is None:
            cache_dir = os.path.join(os.path.expanduser("~"), ".cache", "pyctcdecode")
        return cls.load_from_dir(huggingface_hub.snapshot_download(model_id, cache_dir=cache_dir, **kwargs))

    # -- streaming (decoder.py:669-728) ----------------------------------------------------------
    def get_starting_state(self):
        """decoder.py:669-679: (beams, cached_lm_scores, cached_p_lm_scores) to start a stream with."""
        start_beam = [EMPTY_START_BEAM]
        language_model = self._language_model
        if language_model is None:
            cached_lm_scores: Dict[Any, Any] = {}
        else:
            # (a dict -- with the reference's one starting entry -- that files what device reads return lazily: _LazyMemo)
            cached_lm_scores = _LazyMemo({("", False): (0.0, 0.0, language_model.get_start_state())})
        cached_p_lm_scores: Dict[str, float] = {}
        return start_beam, cached_lm_scores, cached_p_lm_scores

    def _memo_entry(self, cached_lm_scores, text: str):
        """(raw LM score, KenlmState) of a completed-words text; the device returns these for every beam it
        hands out, so a miss only happens for beams the caller built by hand (scored word by word here)."""
        hit = cached_lm_scores.get((text, False))
        if hit is not None:
            return hit[1], hit[2]
        lm = self._language_model
        if ("", False) not in cached_lm_scores:
            cached_lm_scores[("", False)] = (0.0, 0.0, lm.get_start_state())
        _, raw, state = cached_lm_scores[("", False)]
        so_far = ""
        for word in text.split():
            so_far = word if not so_far else so_far + " " + word
            nxt = cached_lm_scores.get((so_far, False))
            if nxt is None:
                score, st = lm.score(state, word)
                nxt = (raw + score, raw + score, st)
                cached_lm_scores[(so_far, False)] = nxt
            _, raw, state = nxt
        return raw, state

    def partial_decode_beams_batch(
        self,
        logits_list: Sequence[Any],
        cached_lm_scores_list: Sequence[Dict[Any, Any]],
        cached_p_lm_scores_list: Sequence[Dict[str, float]],
        beams_list: Sequence[List[Beam]],
        processed_frames_list: Sequence[int],
        beam_width: int = DEFAULT_BEAM_WIDTH,
        beam_prune_logp: float = DEFAULT_PRUNE_LOGP,
        token_min_logp: float = DEFAULT_MIN_TOKEN_LOGP,
        prune_history: bool = DEFAULT_PRUNE_BEAMS,
        hotword_scorer: Optional[HotwordScorer] = None,
        force_next_word: bool = False,
        is_end: bool = False,
    ) -> List[List[LMBeam]]:
        """Many independent streams advanced by one chunk each in ONE device launch (extension of
        partial_decode_beams, decoder.py:681-728).

        The streams are device-resident (ctcdec_stream_*): the beams, their LM states and the words decoded so far stay
        on the GPU between chunks. What comes back is, per stream, a list of LMBeam that fills itself the first time it
        is looked at; handed back unchanged in the next call (the reference's own usage pattern) it never has to, and a
        chunk costs two kernel launches. Lists that are built or edited by the caller -- and everything under
        CTCDEC_RESIDENT_STREAMS=0 -- go the reference's way: the host resolves every beam's strings for the device.
        One caveat of the lazy lists: beams that were handed back can only be read until that next call returns its
        own (the device state has moved on); read them first, or keep ``list(beams)``."""
        with self._call_lock:
            return self._partial_decode_beams_batch_locked(
                logits_list, cached_lm_scores_list, cached_p_lm_scores_list, beams_list, processed_frames_list,
                beam_width, beam_prune_logp, token_min_logp, prune_history, hotword_scorer, force_next_word, is_end)

    def _partial_decode_beams_batch_locked(
        self, logits_list, cached_lm_scores_list, cached_p_lm_scores_list, beams_list, processed_frames_list,
        beam_width, beam_prune_logp, token_min_logp, prune_history, hotword_scorer, force_next_word, is_end,
    ) -> List[List[LMBeam]]:
        n = len(logits_list)
        if n == 0:
            return []
        for logits in logits_list:
            self._check_logits_dimension(logits)
        if hotword_scorer is not None and not isinstance(hotword_scorer, HotwordScorer):
            raise TypeError("hotword_scorer must be a pyctcdecode_amd HotwordScorer")
        weight = hotword_scorer.weight if hotword_scorer is not None else 0.0
        unigrams = hotword_scorer.unigrams if hotword_scorer is not None else []
        key = tuple(unigrams)
        if key != self._hot_key:
            blob, off = B.pack_strings(unigrams)
            self._lib.check(self._lib.dll.ctcdec_set_hotwords(self._handle, blob, B.off_ptr(off), len(unigrams)))
            self._hot_key = key
        for beams in beams_list:
            if not isinstance(beams, _ResidentBeams) and len(beams) == 0:
                raise ValueError("a stream needs at least one beam (use get_starting_state())")
        params = self._params(beam_width, beam_prune_logp, token_min_logp, prune_history, weight, 0)
        # refused here, before any stream is opened, imported into or retired: the lists of the previous chunk stay readable
        # (the library refuses the same values with the same exceptions, api.cpp: decode_impl)
        if params.beam_width < 1:
            raise ValueError("beam_width must be >= 1")
        if params.beam_width > B.MAX_BEAM_WIDTH:
            raise NotImplementedError("beam_width above the supported maximum of %d" % B.MAX_BEAM_WIDTH)
        lazy_ok = os.environ.get("CTCDEC_RESIDENT_STREAMS", "1") != "0"
        # which device streams do these beams belong to?
        streams: Optional[_DeviceStreams] = None
        first = beams_list[0]
        if (isinstance(first, _ResidentBeams) and first._streams.alive(self) and first._streams.n == n and
                first._streams.hot_key == key and
                all(isinstance(b, _ResidentBeams) and b._streams is first._streams and b._index == u and b._current()
                    for u, b in enumerate(beams_list))):
            streams = first._streams  # handed back unchanged: nothing to import
        elif (all(len(b) == 1 and b[0] == EMPTY_START_BEAM for b in beams_list) and
              all(self._memo_starts_at_default(m) for m in cached_lm_scores_list)):
            streams = _DeviceStreams(self, n)  # the reference's starting state: the model's own start state, nothing scored
        else:
            # built or edited by the caller (or fed to another call in between): the host resolves their strings
            streams = _DeviceStreams(self, n)
            streams.import_beams(beams_list, cached_lm_scores_list)
        streams.hot_key = key
        batch = _Batch(logits_list, len(self._idx2vocab))
        if batch.is_device and batch.device_index is not None and batch.device_index != self._device:
            raise ValueError("the logits live on cuda:%d but this decoder was built for cuda:%d (one process per GPU: "
                             "LOCAL_RANK / CTCDEC_DEVICE pick the device)" % (batch.device_index, self._device))
        ptrs, frames = _c_arrays(batch)
        first_frames = (C.c_int32 * n)(*[int(p) for p in processed_frames_list])
        want = bool(is_end) or not lazy_ok
        res = C.c_void_p()
        streams.retire_lists()
        self._lib.check(self._lib.dll.ctcdec_stream_push(
            streams.handle, ptrs, frames, batch.dtype, int(batch.is_device), C.byref(params), first_frames,
            int(bool(force_next_word)), int(bool(is_end)), int(want), C.byref(res)))
        streams.params = params
        streams.memos = list(cached_lm_scores_list)
        try:
            if want:
                return streams.unpack(res)
        finally:
            if res:
                self._lib.dll.ctcdec_result_free(res)
        return streams.lazy_lists()

    def _memo_starts_at_default(self, memo: Dict[Any, Any]) -> bool:
        """Is the memo's entry for the empty text what get_starting_state() puts there -- (0, 0, the model's start state)?
        The reference scores a stream's first word from THAT entry's state (decoder.py:387-396): a caller that seeds it with
        the last_lm_state of a previous segment must not be decoded from the beginning of a sentence (such streams take the
        import path, which reads the entry)."""
        lm = self._language_model
        if lm is None:
            return True
        entry = memo.get(("", False)) if hasattr(memo, "get") else None
        if entry is None:
            return True
        try:
            lm_hw, raw, state = entry
        except (TypeError, ValueError):
            return False
        return lm_hw == 0.0 and raw == 0.0 and _same_lm_state(state, lm.get_start_state())

    def partial_decode_beams(
        self,
        logits: Any,
        cached_lm_scores: Dict[Any, Any],
        cached_p_lm_scores: Dict[str, float],
        beams: List[Beam],
        processed_frames: int,
        beam_width: int = DEFAULT_BEAM_WIDTH,
        beam_prune_logp: float = DEFAULT_PRUNE_LOGP,
        token_min_logp: float = DEFAULT_MIN_TOKEN_LOGP,
        prune_history: bool = DEFAULT_PRUNE_BEAMS,
        hotword_scorer: Optional[HotwordScorer] = None,
        force_next_word: bool = False,
        is_end: bool = False,
    ) -> List[LMBeam]:
        """decoder.py:681-728: advance one stream by one chunk; feed the returned beams back in."""
        return self.partial_decode_beams_batch(
            [logits], [cached_lm_scores], [cached_p_lm_scores], [beams], [processed_frames], beam_width,
            beam_prune_logp, token_min_logp, prune_history, hotword_scorer, force_next_word, is_end,
        )[0]


def build_ctcdecoder(
    labels: List[str],
    kenlm_model_path: Optional[str] = None,
    unigrams: Optional[Collection[str]] = None,
    alpha: float = DEFAULT_ALPHA,
    beta: float = DEFAULT_BETA,
    unk_score_offset: float = DEFAULT_UNK_LOGP_OFFSET,
    lm_score_boundary: bool = DEFAULT_SCORE_LM_BOUNDARY,
) -> BeamSearchDecoderCTC:
    """decoder.py:1051-1099: alphabet from `labels`; with a model path, an n-gram LanguageModel over it. The unigram list
    of an .arpa file is read from the file itself when none is given; any other model format carries none."""
    from_arpa = kenlm_model_path is not None and kenlm_model_path.endswith(".arpa")
    if from_arpa:
        logger.info("Using arpa instead of binary LM file, decoder instantiation might be slow.")
    model = NgramModel(kenlm_model_path) if kenlm_model_path is not None else None
    if unigrams is None and from_arpa:
        unigrams = load_unigram_set_from_arpa(kenlm_model_path)
    elif unigrams is None and model is not None:
        logger.warning("Unigrams not provided and cannot be automatically determined from LM file (only "
                       "arpa format). Decoding accuracy might be reduced.")
    alphabet = Alphabet.build_alphabet(labels)
    if unigrams is not None:
        verify_alphabet_coverage(alphabet, unigrams)
    if model is None:
        return BeamSearchDecoderCTC(alphabet, None)
    lm = LanguageModel(model, unigrams, alpha=alpha, beta=beta, unk_score_offset=unk_score_offset,
                       score_boundary=lm_score_boundary)
    return BeamSearchDecoderCTC(alphabet, lm)
