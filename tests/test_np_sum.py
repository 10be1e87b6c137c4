"""csrc/np_sum.h against numpy itself: the summation order and precision the reference's input sniff depends on
(decoder.py:760, `logits.sum(axis=1).mean()` in the input dtype). The header is compiled into a tiny helper library with
g++ (test infrastructure, tests/_build/); the HIP build runs the same source in utt_sniff_exact."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pyctcdecode_amd", "csrc", "np_sum.h")
OUT_DIR = os.path.join(ROOT, "tests", "_build")
HELPER = r'''
#define CTC_SIM
#include "%s"
extern "C" double np_row_sum_(const void* x, int dtype, long t, long V) { return ctc::np_row_sum(x, dtype, t, V); }
extern "C" double np_mean_(const double* rs, int dtype, long T) { return ctc::np_mean_of_sums(rs, dtype, T); }
extern "C" int np_is_one_(double m) { return ctc::np_mean_is_one(m) ? 1 : 0; }
extern "C" unsigned short to_half_(float f) { return ctc::f32_to_f16_bits(f); }
extern "C" float from_half_(unsigned short h) { return ctc::f16_bits_to_f32(h); }
''' % SRC


@pytest.fixture(scope="module")
def lib():
    os.makedirs(OUT_DIR, exist_ok=True)
    cpp, so = os.path.join(OUT_DIR, "np_sum_probe.cpp"), os.path.join(OUT_DIR, "np_sum_probe.so")
    with open(cpp, "w") as f:
        f.write(HELPER)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-o", so, cpp])
    dll = C.CDLL(so)
    dll.np_row_sum_.restype = C.c_double
    dll.np_row_sum_.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_long]
    dll.np_mean_.restype = C.c_double
    dll.np_mean_.argtypes = [C.c_void_p, C.c_int, C.c_long]
    dll.np_is_one_.argtypes = [C.c_double]
    dll.to_half_.restype = C.c_ushort
    dll.to_half_.argtypes = [C.c_float]
    dll.from_half_.restype = C.c_float
    dll.from_half_.argtypes = [C.c_ushort]
    return dll


@pytest.mark.parametrize("dtype,code", [(np.float32, 0), (np.float64, 1), (np.float16, 2)])
def test_row_sums_and_their_mean_equal_numpy(lib, dtype, code):
    rng = np.random.default_rng(7 + code)
    n_prob = 0
    for V in (1, 5, 7, 8, 9, 29, 32, 100, 127, 128, 129, 257, 1000, 1024, 3000):
        for T in (1, 2, 3, 17, 130, 371):
            x = rng.standard_normal((T, V))
            if rng.random() < 0.6:  # probability rows: the case the sniff exists for
                e = np.exp(x - x.max(axis=1, keepdims=True))
                x = e / e.sum(axis=1, keepdims=True)
            x = np.ascontiguousarray(x.astype(dtype))
            want = x.sum(axis=1)
            got = np.array([lib.np_row_sum_(x.ctypes.data, code, t, V) for t in range(T)])
            assert np.array_equal(got, want.astype(np.float64)), (dtype, V, T)
            mean = lib.np_mean_(got.ctypes.data, code, T)
            assert mean == float(want.mean()), (dtype, V, T, mean, float(want.mean()))
            assert bool(lib.np_is_one_(mean)) == math.isclose(want.mean(), 1)
            n_prob += math.isclose(want.mean(), 1)
    assert n_prob > 10


def test_non_finite_means_are_never_one(lib):
    for m in (float("nan"), float("inf"), -float("inf")):
        assert lib.np_is_one_(m) == 0 and not math.isclose(m, 1)
    assert lib.np_is_one_(1.0) == 1 and lib.np_is_one_(1.0 + 2e-9) == 0 and lib.np_is_one_(1.0 - 5e-10) == 1


def test_half_conversions_equal_numpy(lib):
    for h in range(65536):
        want = np.array([h], dtype=np.uint16).view(np.float16)[0]
        got = lib.from_half_(h)
        assert (np.isnan(want) and np.isnan(got)) or float(want) == got, h
    rng = np.random.default_rng(3)
    with np.errstate(over="ignore"):
        vals = rng.standard_normal(50000).astype(np.float32) * rng.choice([1e-8, 1e-5, 1e-3, 1, 100, 70000], 50000).astype(np.float32)
        want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        assert lib.to_half_(float(v)) == int(w), float(v)
