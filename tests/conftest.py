import os
import sys

# /root/reference is read-only test input: no test process, and no child it starts, may leave bytecode next to its modules
# (tests/test_reference_suite.py and the oracle's differential checks import them from where they lie)
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

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


@pytest.fixture(params=["np", "pk", "f64"])
def both_prune_exps(request, monkeypatch):
    """How the frame-prune kernels treat float32 rows. `np` (the default since round 6): the reference's own float32 arithmetic --
    numpy's float32 exp / log and its pairwise summation order restated (csrc/np_f32.h, np_sum.h) -- exact against the oracle
    fed the float32 array itself (1e-9, strict order). `pk`: round 5's packed float32 polynomial, the float32 bound
    (tests/test_gpu_parity._tol). `f64`: the round-2 fp64 routine, exact against the oracle on the float64 upcast."""
    if request.param == "np":
        monkeypatch.delenv("CTCDEC_PRUNE_EXP", raising=False)
    else:
        monkeypatch.setenv("CTCDEC_PRUNE_EXP", request.param)
    return request.param


def pytest_terminal_summary(terminalreporter):
    """How tight the parity comparisons of this session actually were (tests/golden_util.check_beams)."""
    from tests.golden_util import STATS

    if STATS["beam_lists"]:
        terminalreporter.write_line(
            "parity: %d beam lists / %d beams compared; largest |score - reference| = %.3g (%s); near-tie window used for "
            "%d runs (%d beams)" % (STATS["beam_lists"], STATS["beams"], STATS["max_gap"], STATS["max_gap_what"],
                                    STATS["tie_runs"], STATS["tie_run_beams"]))
        terminalreporter.write_line(
            "strict order: %d of %d beam lists differ from the reference's order (%d of %d runs of EXACTLY equal reference "
            "scores permuted, %d runs of scores within the tie window but not equal)%s" % (
                STATS["lists_order_differs"], STATS["beam_lists"], STATS["exact_tie_runs_permuted"], STATS["exact_tie_runs"],
                STATS["near_tie_runs_permuted"],
                (": " + ", ".join(STATS["lists_order_differs_what"])) if STATS["lists_order_differs_what"] else ""))
