"""TEST INFRASTRUCTURE: build tests/_build/libctcdec_sim.so (sequential CPU execution of
beam_core.h behind the same C ABI).  Never part of the product."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "pyctcdecode_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tests", "_build")
OUT = os.path.join(OUT_DIR, "libctcdec_sim.so")
SOURCES = [os.path.join(SRC, "api.cpp"), os.path.join(SRC, "host_tables.cpp"), os.path.join(SRC, "kenlm_binary.cpp"),
           os.path.join(ROOT, "tests", "sim", "backend_sim.cpp")]
DEPS = SOURCES + [os.path.join(SRC, h) for h in ("common.h", "beam_core.h", "beam_wave.h", "text_wave.h", "set_order.h",
                                                  "np_sum.h", "np_f32.h", "backend.h", "host_tables.h")] + [os.path.join(ROOT, "tests", "sim", "wave_fibers.h"), os.path.join(ROOT, "tests", "sim", "group_fibers.h")] + [os.path.join(ROOT, "include", "ctcdec.h")]


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-DCTC_SIM",
           # (text_wave.h: small odd window / list sizes, so that both end in every possible place of a chain)
           "-DCTC_TEXT_WIN=48", "-DCTC_TEXT_LIST=16", "-Wall", "-Wno-unused-function",
           "-o", OUT] + SOURCES
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
