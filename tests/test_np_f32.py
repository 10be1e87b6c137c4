"""csrc/np_f32.h -- numpy's SIMD float32 exp / log restated -- against numpy itself (the float32 log-softmax of the
reference, decoder.py:180-197, is made of them). The default suite samples 6 M arguments over the ranges a log-softmax and the
probability branch reach plus every special case; `python tools/np_f32_exhaustive.py` compares all 2^32 bit patterns of both
functions (round 6: 0 and 0 differences against numpy 2.2.6 with its AVX512F kernels, profiles/r06_np_f32_exhaustive.txt).
Skipped where numpy's float32 exp is the C library's (no AVX2 / AVX512F: another function, that one is not restated)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SIMD_PROBE = 3776339843435  # sum of the bit patterns of np.exp(linspace(-20, 0, 4001, float32)) under numpy's SIMD kernel


@pytest.fixture(scope="module")
def lib():
    probe = int(np.exp(np.linspace(-20, 0, 4001, dtype=np.float32)).view(np.uint32).astype(np.uint64).sum())
    if probe != SIMD_PROBE:
        pytest.skip("numpy's float32 exp is not its SIMD kernel on this machine (probe %d)" % probe)
    import np_f32_exhaustive

    return C.CDLL(np_f32_exhaustive.helper())


def _run(fn, x):
    y = np.empty_like(x)
    fn(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    return y


def _same(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_exp_equals_numpy(lib):
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.72283, 88.72284, -103.97208, -103.97209, -87.3, -87.4, -100.0, 1e-30,
                        -1e-30, 1e-45, 0.34657359, -0.34657359, 0.34657362, 12582912.0], dtype=np.float32)
    x = np.concatenate([rng.uniform(-104, 0, 2_000_000), rng.uniform(-20, 0, 2_000_000), rng.uniform(0, 89, 500_000),
                        rng.standard_normal(500_000) * 1e-3, special]).astype(np.float32)
    with np.errstate(all="ignore"):
        assert _same(_run(lib.np_exp_arr, x), np.exp(x))


def test_log_equals_numpy(lib):
    rng = np.random.default_rng(12)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, 1.0, 0.70710677, 0.7071068, 0.5, 2.0, 1e-45, 1e-38, 1.17549435e-38,
                        3.4028235e38, 1e-15], dtype=np.float32)
    x = np.concatenate([rng.uniform(1, 1100, 2_000_000), rng.uniform(0, 2, 1_000_000), np.exp(rng.uniform(-87, 88, 1_000_000)),
                        rng.uniform(0, 1e-38, 100_000), special]).astype(np.float32)
    with np.errstate(all="ignore"):
        assert _same(_run(lib.np_log_arr, x), np.log(x))


def test_float32_log_softmax_rows_equal_numpy(lib):
    """The whole row: x - max - log(sum(exp(x - max))) in float32, numpy's pairwise order (np_sum.h) -- checked through the
    simulator's frame prune in tests/test_sim_vs_oracle.py; here the two functions on a row's actual arguments."""
    rng = np.random.default_rng(13)
    for V in (29, 32, 1024, 1025):
        x = (rng.standard_normal((50, V)) * 4).astype(np.float32)
        t = x - x.max(axis=1, keepdims=True)
        e = np.exp(t)
        assert _same(_run(lib.np_exp_arr, np.ascontiguousarray(t.ravel())), e.ravel())
        s = e.sum(axis=1)
        assert _same(_run(lib.np_log_arr, np.ascontiguousarray(s)), np.log(s))
