"""Pin csrc/np_f32.h (numpy's SIMD float32 exp / log, restated) against numpy ITSELF on every float32 bit pattern:
exp on all 2^32 patterns (NaNs, infinities, the overflow / underflow ranges included), log likewise.
Needs a numpy whose float32 exp / log run its SIMD kernels (x86-64 with AVX512F or AVX2 + FMA3: the probe below says which).

    python tools/np_f32_exhaustive.py            (a few minutes on 8 cores; prints the mismatch counts: expected 0 and 0)
"""
import ctypes as C
import os
import subprocess
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHUNK = 1 << 24


def helper():
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    cpp, so = os.path.join(out, "np_f32_probe.cpp"), os.path.join(out, "np_f32_probe.so")
    with open(cpp, "w") as f:
        f.write('#define CTC_SIM\n#include "%s"\n' % os.path.join(ROOT, "pyctcdecode_amd", "csrc", "np_f32.h") +
                'extern "C" void np_exp_arr(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = ctc::np_exp_f32(x[i]); }\n'
                'extern "C" void np_log_arr(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = ctc::np_log_f32(x[i]); }\n')
    # (-mfma -ffp-contract=fast ON PURPOSE: the header itself has to keep the compiler from fusing what numpy does not fuse)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=fast", "-shared", "-fPIC", "-o", so, cpp])
    return so


def chunk(args):
    so, start = args
    dll = C.CDLL(so)
    x = np.arange(start, start + CHUNK, dtype=np.uint64).astype(np.uint32).view(np.float32)
    y = np.empty_like(x)
    bad = []
    for name, ref in (("np_exp_arr", np.exp), ("np_log_arr", np.log)):
        getattr(dll, name)(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
        with np.errstate(all="ignore"):
            want = ref(x)
        w, g = want.view(np.uint32), y.view(np.uint32)
        differs = (w != g) & ~(np.isnan(want) & np.isnan(y))  # (any NaN equals any NaN: payloads are not part of the contract)
        bad.append((int(differs.sum()), [hex(int(v)) for v in x.view(np.uint32)[differs][:4]]))
    return start, bad


def main():
    so = helper()
    probe = np.exp(np.linspace(-20, 0, 4001, dtype=np.float32)).view(np.uint32).astype(np.uint64).sum()
    print("numpy %s, float32 exp probe sum %d (3776339843435 = the SIMD kernel, 3776339843710 = the C library's expf)" % (np.__version__, probe))
    tot = [0, 0]
    first = [[], []]
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for start, bad in ex.map(chunk, [(so, s) for s in range(0, 1 << 32, CHUNK)]):
            for k in range(2):
                tot[k] += bad[k][0]
                if bad[k][0] and len(first[k]) < 8:
                    first[k] += bad[k][1]
    print("exp: %d of 2^32 bit patterns differ from numpy %s" % (tot[0], first[0] or ""))
    print("log: %d of 2^32 bit patterns differ from numpy %s" % (tot[1], first[1] or ""))
    return 0 if tot == [0, 0] else 1


if __name__ == "__main__":
    sys.exit(main())
