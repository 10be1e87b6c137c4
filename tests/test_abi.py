"""CPU: the C-ABI library loads and exports every symbol include/ctcdec.h declares (no compute
calls: there is no GPU here), and the product path fails loudly without a device / library."""
import os
import re

import pytest

from pyctcdecode_amd import _binding as B
from pyctcdecode_amd import build as build_mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return build_mod.build(verbose=False)


def test_header_symbols_all_exported(lib_path):
    with open(os.path.join(ROOT, "include", "ctcdec.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(ctcdec_[a-z0-9_]+)\s*\(", header))
    assert declared == set(B.EXPORTED_SYMBOLS), declared ^ set(B.EXPORTED_SYMBOLS)
    lib = B.Library(lib_path)  # raises AttributeError on a missing export
    assert lib.dll.ctcdec_version().startswith(b"ctcdec")


def test_product_library_is_the_hip_build(lib_path):
    import subprocess

    out = subprocess.run(["ldd", lib_path], capture_output=True, text=True).stdout
    assert "libamdhip64" in out, out
    syms = subprocess.run(["nm", "-D", lib_path], capture_output=True, text=True).stdout
    assert "hipLaunchKernel" in syms or "__hipRegisterFatBinary" in syms


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError):
        B.Library(str(tmp_path / "nope.so"))


def test_no_gpu_means_error_not_fallback(lib_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyctcdecode_amd import build_ctcdecoder

    with pytest.raises(B.NativeError):
        build_ctcdecoder([" ", "a", "b"])


def test_product_never_imports_oracle_or_sim():
    pkg = os.path.join(ROOT, "pyctcdecode_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "import oracle" not in src and "from oracle" not in src, fn
                assert "libctcdec_sim" not in src and "backend_sim" not in src.replace(
                    "tests/sim/backend_sim.cpp", ""), fn


def _documented_build_commands():
    """The by-hand build of INTEGRATION.md's "## Build" block: every command line after the `# or by hand` comment."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    block = doc[doc.index("## Build"):]
    block = block[block.index("```bash") + len("```bash"):]
    block = block[:block.index("```")]
    lines = [ln.strip() for ln in block.splitlines()]
    by_hand = lines[[i for i, ln in enumerate(lines) if ln.startswith("# or by hand")][0] + 1:]
    return [ln for ln in by_hand if ln and not ln.startswith("#")]


def test_documented_by_hand_build_links_and_exports_everything(tmp_path):
    """INTEGRATION.md's by-hand build is run as written (cwd = a scratch directory, sources by absolute path, the library
    into the scratch directory): it has to name every translation unit and flag pyctcdecode_amd/build.py uses, and what
    it produces has to export the whole C ABI. (Round 4's text had fallen behind the build: it no longer linked.)"""
    import shlex
    import subprocess
    from concurrent.futures import ThreadPoolExecutor

    cmds = [shlex.split(c) for c in _documented_build_commands()]
    assert len(cmds) == len(build_mod.SOURCES) + 1, cmds
    out = str(tmp_path / "libctcdec.so")

    def localise(cmd):
        cmd = [a if not a.startswith("pyctcdecode_amd/csrc/") else os.path.join(ROOT, a) for a in cmd]
        cmd = [out if a == "pyctcdecode_amd/libctcdec.so" else a for a in cmd]
        if cmd[0] == "hipcc":
            cmd[0] = build_mod.hipcc()
        return cmd

    compiles, link = [localise(c) for c in cmds[:-1]], localise(cmds[-1])
    # the documented lines and build.py agree on sources and on the per-file flags
    for src in build_mod.SOURCES:
        mine = [c for c in compiles if c[c.index("-c") + 1].endswith("/" + src)]
        assert len(mine) == 1, src
        for flag in build_mod.HIP_FLAGS.get(src, []):
            assert flag in mine[0], (src, flag)
        if src.endswith(".hip"):
            assert "--offload-arch=gfx950" in mine[0] and "-O3" in mine[0], mine[0]

    def run(cmd):
        r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True)
        assert r.returncode == 0, (" ".join(cmd), r.stderr[-2000:])

    with ThreadPoolExecutor(len(compiles)) as ex:
        list(ex.map(run, compiles))
    run(link)
    lib = B.Library(out)  # raises AttributeError on a missing export
    assert lib.dll.ctcdec_version().startswith(b"ctcdec")


def test_python_limits_match_the_header():
    with open(os.path.join(ROOT, "include", "ctcdec.h")) as f:
        header = f.read()
    assert int(re.search(r"#define\s+CTCDEC_MAX_BEAM_WIDTH\s+(\d+)", header).group(1)) == B.MAX_BEAM_WIDTH


def test_wave_kernel_keeps_its_register_budget():
    """Four waves per SIMD -- sixteen utterances per CU, the headline batch in one round -- need the wave kernel within 128
    registers WITHOUT spills to scratch memory; that rests on an internal LLVM switch (-mllvm -disable-machine-licm, build.py)
    that a toolchain upgrade could silently change. The compiler's own resource remarks for the shipped flags are the check."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "beam_wave<100, 4, false"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1000:]
    rows = [ln.split() for ln in out.stdout.splitlines() if "beam_wave<100, 4, false>" in ln]
    assert len(rows) == 1, out.stdout
    sgpr, vgpr, agpr, scratch, sgpr_spill, vgpr_spill, occ = (int(v) for v in rows[0][-8:-1])
    assert vgpr + agpr <= 128 and vgpr_spill == 0 and occ == 4, rows[0]


def test_committed_pmc_summary_is_bound_to_the_tree_it_was_measured_on(monkeypatch):
    """bench.py fills roofline.traffic only from a PMC summary of THIS library: by the file's hash, or -- a rebuild of the same tree
    under another root differs in hipcc's path-derived symbols -- by the hash of sources, command lines and compilers."""
    import argparse
    import json
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    tag = build_mod.source_tag()
    assert build_mod.source_tag() == tag and len(tag) == 64
    monkeypatch.setenv("CTCDEC_HIPCC_EXTRA", "-DSOMETHING_ELSE")
    assert build_mod.source_tag() != tag  # other flags are another library
    monkeypatch.delenv("CTCDEC_HIPCC_EXTRA")
    with open(os.path.join(root, bench.PMC_FILE)) as f:
        pmc = json.load(f)
    args = argparse.Namespace(frames=bench.T, workload="headline")
    got, why = bench.load_pmc(args)
    if pmc["source_sha16"] == tag[:16] and build_mod._lib_stamp_ok():
        assert got is not None and "bound_by" in got, why
    else:  # the tree moved on after the passes: the summary must be refused, loudly
        assert got is None and "traffic withheld" in why
    monkeypatch.setattr(bench, "binary_tag", lambda: "0" * 16)
    monkeypatch.setattr(build_mod, "source_tag", lambda: "1" * 64)
    got, why = bench.load_pmc(args)
    assert got is None and "traffic withheld" in why
