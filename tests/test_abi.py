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
