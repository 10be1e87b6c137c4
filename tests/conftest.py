import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(params=["wave", "group"])
def both_beam_kernels(request, monkeypatch):
    """Run a parity test once per beam kernel: `wave` = one wavefront per utterance (csrc/beam_wave.h, used whenever
    the decode is eligible for it), `group` = one workgroup per utterance (csrc/beam_core.h). Without this the
    library picks by batch size, and small test batches would only ever see the workgroup kernel."""
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", request.param)
    return request.param
