"""csrc/set_order_small.h (the 48-slot tables the fast frame-prune kernel orders 64 frames at once with) against
csrc/set_order.h and against real Python sets: every key count it accepts, argmax inside and outside the ids. The header is
compiled into a tiny helper library with g++ (test infrastructure, tests/_build/)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyctcdecode_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tests", "_build")
HELPER = r'''
#define CTC_SIM
#include "%s/set_order.h"
#include "%s/set_order_small.h"
struct Tab {
  uint16_t s[48];
  uint16_t get(uint32_t k) const { return s[k]; }
  void put(uint32_t k, uint16_t v) { s[k] = v; }
};
// out: ids in iteration order, pay: their payloads; returns the count
extern "C" int small_order_(const uint16_t* ids, int n, int argmax, uint16_t* out, uint16_t* pay) {
  Tab t;
  for (int k = 0; k < 48; ++k) t.s[k] = 0x1234;  // (stale contents must not matter)
  ctc::SmallSet r = ctc::small_set_order(t, (uint32_t)n, [ids](uint32_t k) { return (uint32_t)ids[k]; }, (uint32_t)argmax);
  int m = 0;
  for (uint32_t k = 0; k <= r.mask; ++k) {
    const uint16_t v = t.get(r.base + k);
    if (v != ctc::SMALL_SET_EMPTY) {
      out[m] = v & ctc::SMALL_SET_ID_MASK;
      pay[m] = v >> ctc::SMALL_SET_ID_BITS;
      ++m;
    }
  }
  return m == (int)r.used ? m : -1;
}
extern "C" int big_order_(const uint16_t* ids, int n, int argmax, uint16_t* out) {
  uint16_t a[256], r[256], sc[256];
  return (int)ctc::cpython_set_order(ids, (uint32_t)n, (uint32_t)argmax, a, r, sc, out);
}
''' % (CSRC, CSRC)


@pytest.fixture(scope="module")
def lib():
    os.makedirs(OUT_DIR, exist_ok=True)
    cpp, so = os.path.join(OUT_DIR, "set_small_probe.cpp"), os.path.join(OUT_DIR, "set_small_probe.so")
    with open(cpp, "w") as f:
        f.write(HELPER)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-o", so, cpp])
    dll = C.CDLL(so)
    dll.small_order_.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    dll.big_order_.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return dll


def _small(lib, ids, amax):
    a = np.asarray(ids, dtype=np.uint16)
    out, pay = np.zeros(17, np.uint16), np.zeros(17, np.uint16)
    m = lib.small_order_(a.ctypes.data, len(ids), int(amax), out.ctypes.data, pay.ctypes.data)
    assert m >= 0
    return [int(v) for v in out[:m]], [int(v) for v in pay[:m]]


def _big(lib, ids, amax):
    a = np.asarray(ids, dtype=np.uint16)
    out = np.zeros(40, np.uint16)
    m = lib.big_order_(a.ctypes.data, len(ids), int(amax), out.ctypes.data)
    return [int(v) for v in out[:m]]


def test_small_tables_give_the_order_of_a_real_set(lib):
    rng = np.random.default_rng(5)
    n_cases = 0
    for n in range(0, 16):
        for rep in range(400):
            hi = [1024, 4095, 64, 40, 300, 2048][rep % 6]  # dense id ranges collide in the 8-slot table; ids go up to 4094
            ids = sorted(int(v) for v in rng.choice(hi, size=min(n, hi), replace=False))
            inside = n > 0 and rep % 3 != 0
            amax = int(rng.choice(ids)) if inside else int(rng.integers(0, hi))
            got, pay = _small(lib, ids, amax)
            real = [int(k) for k in (set(ids) | {amax})]  # decoder.py:445-447
            assert got == real, (ids, amax)
            assert got == _big(lib, ids, amax)
            for g, p in zip(got, pay):  # payloads: the index of the id, 15 for an argmax from outside
                assert (p == 15 and g == amax and amax not in ids) or ids[p] == g
            n_cases += 1
    assert n_cases == 16 * 400


def test_small_tables_on_runs_of_neighbouring_ids(lib):
    """Consecutive ids fill the linear-probe windows: the case the window arithmetic exists for."""
    for start in (0, 1, 7, 8, 25, 31, 500, 1009):
        for n in range(0, 16):
            if start + n > 1024:
                continue
            ids = list(range(start, start + n))
            for amax in (0, start, start + n - 1 if n else 3, (start + 8) % 1024, (start + 32) % 1024, 1023, 2047, 4094):
                got, _ = _small(lib, ids, amax)
                assert got == [int(k) for k in (set(ids) | {amax})], (ids, amax)
