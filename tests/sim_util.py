"""TEST INFRASTRUCTURE: route the Python shell to the sequential CPU build of beam_core.h
(tests/_build/libctcdec_sim.so) so that host logic + beam logic can be checked without a GPU.
Only tests call this; the product always loads pyctcdecode_amd/libctcdec.so.
CTCDEC_SIM_LIB=<path> substitutes another build of the simulator, e.g. the AddressSanitizer / UBSan one of
tools/sim_sanitized.sh."""
import os

import pytest

from pyctcdecode_amd import _binding as B


@pytest.fixture()
def sim_library(monkeypatch):
    from tests.sim.build_sim import build

    lib = B.Library(os.environ.get("CTCDEC_SIM_LIB") or build())
    monkeypatch.setattr(B, "_LIB", lib)
    return lib
