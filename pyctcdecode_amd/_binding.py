"""ctypes binding of include/ctcdec.h.

The library is the in-tree ``pyctcdecode_amd/libctcdec.so`` built by ``__graft_entry__.build()``
(hipcc, gfx950).  There is no fallback of any kind: if the shared object is missing or no HIP
device is usable, every decode entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctcdec.so")

MAX_CTX = 5


class Params(C.Structure):
    _fields_ = [
        ("beam_width", C.c_int32),
        ("prune_history", C.c_int32),
        ("n_best", C.c_int32),
        ("want_lm_state", C.c_int32),
        ("beam_prune_logp", C.c_double),
        ("token_min_logp", C.c_double),
        ("hotword_weight", C.c_double),
        ("alpha", C.c_double),
        ("beta", C.c_double),
        ("unk_score_offset", C.c_double),
        ("log_base_change", C.c_double),
        ("lm_score_boundary", C.c_int32),
        ("first_frame", C.c_int32),
        ("texts_only", C.c_int32),
        ("reserved", C.c_int32),
    ]


class LmState(C.Structure):
    _fields_ = [("length", C.c_int32), ("words", C.c_uint32 * MAX_CTX), ("backoff", C.c_float * MAX_CTX)]


class Packed(C.Structure):
    _fields_ = [
        ("n_utts", C.c_int64),
        ("n_beams", C.c_int64),
        ("n_words", C.c_int64),
        ("beam_off", C.POINTER(C.c_int64)),
        ("text_blob", C.c_void_p),
        ("text_off", C.POINTER(C.c_int64)),
        ("logit_score", C.POINTER(C.c_double)),
        ("lm_score", C.POINTER(C.c_double)),
        ("word_cnt_off", C.POINTER(C.c_int64)),
        ("word_byte_off", C.POINTER(C.c_int32)),
        ("word_start", C.POINTER(C.c_int32)),
        ("word_end", C.POINTER(C.c_int32)),
        ("lm_state", C.POINTER(LmState)),
        ("partial_blob", C.c_void_p),
        ("partial_off", C.POINTER(C.c_int64)),
        ("src_beam", C.POINTER(C.c_int32)),
        ("last_char", C.POINTER(C.c_int32)),
        ("partial_start", C.POINTER(C.c_int32)),
        ("partial_end", C.POINTER(C.c_int32)),
        ("raw_lm_score", C.POINTER(C.c_double)),
    ]


class BeamIn(C.Structure):
    _fields_ = [
        ("logit_score", C.c_double),
        ("raw_lm_score", C.c_double),
        ("lm_state", LmState),
        ("last_char", C.c_int32),
        ("partial_start", C.c_int32),
        ("partial_end_frame", C.c_int32),
        ("reserved", C.c_int32),
        ("text_begin", C.c_int64),
        ("text_end", C.c_int64),
        ("partial_begin", C.c_int64),
        ("partial_end", C.c_int64),
        ("more_states", C.POINTER(LmState)),
    ]


# every symbol include/ctcdec.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_PROTOS = {
    "ctcdec_create": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "ctcdec_destroy": (None, [_VP]),
    "ctcdec_lm_load_arpa": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int32)]),
    "ctcdec_lm_save_flat": (C.c_int, [_VP, C.c_char_p]),
    "ctcdec_lm_load_flat": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int32)]),
    "ctcdec_lm_load_kenlm": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int32)]),
    "ctcdec_is_kenlm_binary": (C.c_int, [C.c_char_p]),
    "ctcdec_arpa_to_kenlm_binary": (C.c_int, [C.c_char_p, C.c_char_p, C.c_float]),
    "ctcdec_lm_set_unigrams": (C.c_int, [_VP, C.c_int32, C.c_char_p, C.POINTER(C.c_int64), C.c_int64,
                                         C.POINTER(C.c_int64)]),
    "ctcdec_lm_share": (C.c_int, [_VP, _VP]),
    "ctcdec_lm_clone": (C.c_int, [_VP, _VP]),
    "ctcdec_lm_share_multi": (C.c_int, [_VP, C.POINTER(_VP), C.c_int32]),
    "ctcdec_lm_set_params": (C.c_int, [_VP, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32]),
    "ctcdec_lm_count": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "ctcdec_lm_prefix_flags": (C.c_int, [_VP, C.c_char_p, C.c_int64, C.POINTER(C.c_uint32)]),
    "ctcdec_lm_word_index": (C.c_int, [_VP, C.c_char_p, C.c_int64, C.POINTER(C.c_uint32)]),
    "ctcdec_lm_word_string": (C.c_int, [_VP, C.c_uint32, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "ctcdec_lm_start_state": (C.c_int, [_VP, C.c_int32, C.POINTER(LmState)]),
    "ctcdec_lm_base_score": (C.c_int, [_VP, C.POINTER(LmState), C.c_uint32, C.POINTER(LmState),
                                       C.POINTER(C.c_float)]),
    "ctcdec_set_hotwords": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int64), C.c_int64]),
    "ctcdec_decode_batch": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(Params), C.POINTER(LmState), C.POINTER(_VP)]),
    "ctcdec_decode_stream_batch": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32,
                                             C.POINTER(Params), C.POINTER(C.c_int32), C.POINTER(BeamIn),
                                             C.POINTER(C.c_int64), C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "ctcdec_stream_open": (C.c_int, [_VP, C.c_int32, C.POINTER(LmState), C.POINTER(_VP)]),
    "ctcdec_stream_push": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(Params),
                                     C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "ctcdec_stream_read": (C.c_int, [_VP, C.POINTER(Params), C.POINTER(_VP)]),
    "ctcdec_stream_import": (C.c_int, [_VP, C.POINTER(BeamIn), C.POINTER(C.c_int64), C.c_char_p, C.c_int64]),
    "ctcdec_stream_frames": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "ctcdec_stream_close": (None, [_VP]),
    "ctcdec_result_num_utts": (C.c_int32, [_VP]),
    "ctcdec_result_num_beams": (C.c_int32, [_VP, C.c_int32]),
    "ctcdec_result_text": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "ctcdec_result_scores": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ctcdec_result_frames": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                       C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)),
                                       C.POINTER(C.POINTER(C.c_int32))]),
    "ctcdec_result_lm_state": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(LmState)]),
    "ctcdec_frame_survivors": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "ctcdec_result_lm_state_of": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(LmState)]),
    "ctcdec_result_pack": (C.c_int, [_VP, C.POINTER(Packed)]),
    "ctcdec_result_texts": (C.c_int, [_VP, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]),
    "ctcdec_result_texts_joined": (C.c_int, [_VP, C.c_char, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ctcdec_result_text_blocks": (C.c_int, [_VP, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_int64)),
                                           C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]),
    "ctcdec_result_timing": (C.c_int, [_VP, C.POINTER(C.c_double)]),
    "ctcdec_result_beam_kernel": (C.c_int, [_VP]),
    "ctcdec_device": (C.c_int, []),
    "ctcdec_result_free": (None, [_VP]),
    "ctcdec_profile_phases": (C.c_int, [_VP, C.c_int32, C.POINTER(C.c_uint64), C.c_int32]),
    "ctcdec_last_error": (C.c_char_p, []),
    "ctcdec_version": (C.c_char_p, []),
}
MAX_BEAM_WIDTH = 256  # CTCDEC_MAX_BEAM_WIDTH of include/ctcdec.h (tests/test_abi.py holds the two together)
EXPORTED_SYMBOLS = tuple(_PROTOS)


class NativeError(RuntimeError):
    pass


class Library:
    """A loaded libctcdec with typed prototypes."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(
                "%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()'). pyctcdecode_amd has no CPU fallback." % path
            )
        self.path = path
        # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as the system
        # one). If this library were loaded first it would pull in /opt/rocm's copy and torch's later
        # initialisation would find "no HIP GPUs"; importing torch first makes both share torch's copy,
        # which is also what makes torch device pointers valid in our kernels.
        try:
            import torch  # noqa: F401
        except ImportError:  # standalone use: the system HIP runtime
            pass
        self.dll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.dll, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args

    def check(self, rc: int) -> None:
        if rc == 0:
            return
        msg = (self.dll.ctcdec_last_error() or b"").decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        if rc == -2:
            raise OSError(msg)
        if rc == -4:
            raise NotImplementedError(msg)
        raise NativeError("libctcdec error %d: %s" % (rc, msg))


_LIB: Optional[Library] = None
_PYTEXTS = None


def _pytexts():
    global _PYTEXTS
    if _PYTEXTS is None:
        path = os.path.join(_HERE, "_pytexts.so")
        _PYTEXTS = False
        if os.path.exists(path):
            try:
                dll = C.PyDLL(path)  # (keeps the GIL: the functions build Python objects)
                dll.ctcdec_py_split_texts.restype = C.py_object
                dll.ctcdec_py_split_texts.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_char]
                dll.ctcdec_py_texts_from_blocks.restype = C.py_object
                dll.ctcdec_py_texts_from_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int64]
                dll.ctcdec_py_output_beams.restype = C.py_object
                dll.ctcdec_py_output_beams.argtypes = [C.py_object, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.py_object, C.py_object]
                dll.ctcdec_py_lm_beams.restype = C.py_object
                dll.ctcdec_py_lm_beams.argtypes = [C.py_object, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                                                   C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int32), C.py_object,
                                                   C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                   C.POINTER(C.c_double), C.py_object]
                _PYTEXTS = dll
            except (OSError, AttributeError):
                _PYTEXTS = False
    return _PYTEXTS


def texts_of(lib: "Library", res) -> Optional[list]:
    """All texts of a result as a list of str, one PyUnicode_DecodeUTF8 per text straight from the library's memory
    (csrc/pytexts.c); None when that helper was not built (the caller then joins and splits)."""
    dll = _pytexts()
    if not dll:
        return None
    pool, off, ln, n = C.c_void_p(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_int64()
    lib.check(lib.dll.ctcdec_result_text_blocks(res, C.byref(pool), C.byref(off), C.byref(ln), C.byref(n)))
    if n.value == 0:
        return []
    return dll.ctcdec_py_texts_from_blocks(pool, off, ln, n.value)


def output_beams(cls, pk: "Packed", states: Optional[list], frames_of=None) -> Optional[list]:
    """The OutputBeam lists of a packed result, built in C (csrc/pytexts.c); None when the helper was not built.
    frames_of(text, w0, w1): the caller's (lazy) text_frames object for words w0 .. w1-1 of the result."""
    dll = _pytexts()
    if not dll:
        return None
    return dll.ctcdec_py_output_beams(cls, pk.n_utts, pk.beam_off, pk.text_off, pk.text_blob, pk.logit_score, pk.lm_score,
                                      pk.word_cnt_off, pk.word_start, pk.word_end, states, frames_of)


def lm_beams(cls, n_streams: int, pk: "Packed", labels: list, frames_of=None) -> Optional[list]:
    """The LMBeam lists of a packed streaming result, built in C (csrc/pytexts.c); None when the helper was not built.
    frames_of(w0, w1): the caller's (lazy) text_frames object for words w0 .. w1-1 of the result, instead of eager lists."""
    dll = _pytexts()
    if not dll:
        return None
    return dll.ctcdec_py_lm_beams(cls, n_streams, pk.beam_off, pk.text_off, pk.text_blob, pk.partial_off, pk.partial_blob,
                                  pk.last_char, labels, pk.word_cnt_off, pk.word_start, pk.word_end, pk.partial_start,
                                  pk.partial_end, pk.logit_score, pk.lm_score, frames_of)


def split_texts(blob_ptr, nbytes: int, n: int, sep: bytes):
    """n UTF-8 texts separated by `sep` in the library's memory -> list of str (csrc/pytexts.c when it was built, else in
    Python)."""
    dll = _pytexts()
    if dll:
        return dll.ctcdec_py_split_texts(blob_ptr, nbytes, n, sep)
    return C.string_at(blob_ptr, nbytes).decode("utf-8").split(sep.decode("ascii"))


def get_library() -> Library:
    global _LIB
    if _LIB is None:
        _LIB = Library()
    return _LIB


def pack_strings(strings: Sequence[str]):
    """UTF-8 blob + int64 offsets (n+1)."""
    enc = [s.encode("utf-8") for s in strings]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        off[1:] = np.cumsum([len(e) for e in enc])
    blob = b"".join(enc)
    return blob, off


def off_ptr(off: np.ndarray):
    return off.ctypes.data_as(C.POINTER(C.c_int64))
