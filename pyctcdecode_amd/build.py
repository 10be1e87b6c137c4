"""Build pyctcdecode_amd/libctcdec.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libctcdec.so")
SOURCES = ["api.cpp", "host_tables.cpp", "kenlm_binary.cpp", "backend_hip.hip", "beam_wave_hip.hip", "beam_group_hip.hip"]
# The wave kernel is one loop over the frames with ~18 000 instructions in its body and a budget of 128 registers. LLVM's
# machine-level loop-invariant code motion hoists every constant and address computation it finds out of that loop and
# keeps them in registers across it: 105 registers spilled to scratch memory (and every reload of one waits for the
# stores in flight). Without the pass: none.
# Round 5: the machine scheduler's "max-ilp" strategy for the frame-prune and wave kernels. Their occupancy is fixed by LDS and by
# the waves-per-SIMD attribute, so the default strategy's goal (the fewest registers, loads issued right before their use) buys
# nothing; scheduling for instruction-level parallelism took 6 % off frame_prune_fast (4.22 -> 3.98 ms, A/B in one process) and
# 1.4 % off a lone wave's frame; at sixteen waves per CU the wave kernel is unchanged (profiles/r05_ab_experiments.log).
_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
HIP_FLAGS = {"backend_hip.hip": _MAX_ILP,
             "beam_wave_hip.hip": ["-mllvm", "-disable-machine-licm"] + _MAX_ILP,
             # (the workgroup kernel: 238 -> 195 registers, 255 -> 138 scalar registers spilled to vector lanes)
             "beam_group_hip.hip": ["-mllvm", "-disable-machine-licm"]}
HEADERS = ["common.h", "beam_core.h", "beam_wave.h", "set_order.h", "set_order_small.h", "backend.h", "host_tables.h", "np_sum.h",
           "np_f32.h", "wave_ops_hip.h", "text_wave.h"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


_VERSIONS = {}


def _tool_version(tool: str) -> str:
    """First lines of `<tool> --version` (cached): part of every object's stamp, so that another compiler rebuilds."""
    if tool not in _VERSIONS:
        try:
            out = subprocess.run([tool, "--version"], capture_output=True, text=True).stdout
        except OSError:
            out = "?"
        _VERSIONS[tool] = " ".join(out.split()[:40])
    return _VERSIONS[tool]


def _compile_cmd(src: str, obj: str):
    path = os.path.join(SRC, src)
    if src.endswith(".hip"):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-Wno-unused-result"]
        return cmd + HIP_FLAGS.get(src, []) + os.environ.get("CTCDEC_HIPCC_EXTRA", "").split() + ["-c", path, "-o", obj]
    # pure host C++ (no HIP headers): ARPA parsing, table builders, C ABI
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    return [cxx, "-O2", "-std=c++17", "-fPIC", "-DNDEBUG", "-c", path, "-o", obj]


def _stamp(cmd) -> str:
    """What an object was built with: its full command line and the compiler's version. An object whose stamp differs
    from what build() would run now is stale whatever its mtime says (changed flags, CTCDEC_HIPCC_EXTRA, CXX, a new ROCm)."""
    import hashlib

    # (paths relative to the package: the same tree under another root -- the GPU box's scratch copy -- is the same build)
    line = " ".join(cmd).replace(HERE, "$PKG")
    return hashlib.sha256((line + "\n" + _tool_version(cmd[0])).encode()).hexdigest()


def _stamp_ok(obj: str, cmd) -> bool:
    try:
        with open(obj + ".stamp") as f:
            return f.read().strip() == _stamp(cmd)
    except OSError:
        return False


def _obj_dir() -> str:
    return os.path.join(HERE, "csrc", "_obj")


def source_tag() -> str:
    """SHA-256 over what the library is made from: every source and header (name and bytes), every translation unit's command line
    (paths relative to the package) and the compilers' versions. Two trees with the same tag build the same kernels; the library
    FILE is not that stable: hipcc derives each translation unit's `__hip_cuid_*` symbol from the absolute path it compiles."""
    import hashlib

    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS) + ["../../include/ctcdec.h"]:
        with open(os.path.join(SRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    for src in SOURCES:
        cmd = _compile_cmd(src, os.path.join(_obj_dir(), src + ".o"))
        # (the compiler by its name and version, not by where this machine keeps it)
        line = " ".join([os.path.basename(cmd[0])] + cmd[1:]).replace(HERE, "$PKG")
        h.update((line + "\n" + _tool_version(cmd[0]) + "\n").encode())
    return h.hexdigest()


def _lib_stamp_ok():
    """True / False when libctcdec.so.stamp says the library is / is not built from this tree, None without a stamp.
    A box that ships the library and its stamp but no hipcc cannot rebuild anything: what is there counts as built."""
    try:
        with open(OUT + ".stamp") as f:
            stamp = f.read().strip()
    except OSError:
        return None
    try:
        return stamp == source_tag()
    except RuntimeError:  # hipcc not found
        return True


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    ok = _lib_stamp_ok()
    if ok is not None:
        return not ok  # (contents, not mtimes: a checkout that refreshes every file's time does not rebuild the same library)
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC, f) for f in SOURCES + HEADERS + ["pytexts.c"]] + [os.path.join(HERE, "..", "include", "ctcdec.h")]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    # objects of another command line / compiler (only where objects exist: a tree that shipped just the library is as built)
    for src in SOURCES:
        obj = os.path.join(_obj_dir(), src + ".o")
        if os.path.exists(obj) and not _stamp_ok(obj, _compile_cmd(src, obj)):
            return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        if pytexts_needs_build():
            build_pytexts(verbose)
        return OUT
    obj_dir = _obj_dir()
    os.makedirs(obj_dir, exist_ok=True)
    deps = [os.path.join(SRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "ctcdec.h")]
    newest_header = max(os.path.getmtime(d) for d in deps)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src + ".o")
        objs.append(obj)
        cmd = _compile_cmd(src, obj)
        if (not force and os.path.exists(obj) and _stamp_ok(obj, cmd) and
                os.path.getmtime(obj) >= max(newest_header, os.path.getmtime(os.path.join(SRC, src)))):
            continue  # this translation unit is up to date: same sources, same command line, same compiler
        jobs.append(cmd)
    from concurrent.futures import ThreadPoolExecutor

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(cmd[-1] + ".stamp", "w") as f:
            f.write(_stamp(cmd))

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))  # (the translation units compile side by side; an error of any of them is raised here)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    with open(OUT + ".stamp", "w") as f:
        f.write(source_tag())
    build_pytexts(verbose)
    return OUT


PYTEXTS = os.path.join(HERE, "_pytexts.so")


def _pytexts_cmd():
    import sysconfig

    inc = sysconfig.get_paths().get("include")
    if not inc or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    cc = os.environ.get("CC") or shutil.which("gcc") or "gcc"
    return [cc, "-O2", "-fPIC", "-shared", "-I" + inc, os.path.join(SRC, "pytexts.c"), "-o", PYTEXTS + ".tmp"]


def _pytexts_tag(cmd) -> str:
    """The helper's own stamp (the library's source_tag() does not cover it): its source bytes, command line and compiler."""
    import hashlib

    with open(os.path.join(SRC, "pytexts.c"), "rb") as f:
        body = f.read()
    return hashlib.sha256(body + b"\0" + _stamp(cmd).encode()).hexdigest()


def pytexts_needs_build() -> bool:
    cmd = _pytexts_cmd()
    if cmd is None:
        return False  # cannot be built here (no Python headers): the shell splits the blob in Python
    if not os.path.exists(PYTEXTS):
        return True
    try:
        with open(PYTEXTS + ".stamp") as f:
            return f.read().strip() != _pytexts_tag(cmd)
    except OSError:
        return os.path.getmtime(os.path.join(SRC, "pytexts.c")) > os.path.getmtime(PYTEXTS)


def build_pytexts(verbose: bool = True) -> str:
    """The shell's one C helper (csrc/pytexts.c: a result's texts as a list of str without three passes over the blob).
    Optional: without Python's headers the shell splits the blob in Python."""
    cmd = _pytexts_cmd()
    if cmd is None:
        return ""
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError):
        return ""
    with open(PYTEXTS + ".stamp", "w") as f:
        f.write(_pytexts_tag(cmd))
    os.replace(PYTEXTS + ".tmp", PYTEXTS)
    return PYTEXTS


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
