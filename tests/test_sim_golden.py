"""CPU: the Python shell + host tables + beam_core.h (run sequentially by the test-only sim
backend) against the reference's golden vectors.  This checks the LOGIC the HIP kernels compile;
the kernels themselves are checked by tests/test_gpu_parity.py on a GPU."""
import numpy as np
import pytest

from tests.golden_util import check_beams, lm_path, load_cases
from tests.sim_util import sim_library  # noqa: F401

pytestmark = pytest.mark.usefixtures("both_beam_kernels")

CASES, INPUTS = load_cases()


def run_case(case):
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(case["labels"], lm_path(case["lm"]), case["unigrams"], **case["build"])
    x = INPUTS[case["input"]]
    out = dec.decode_beams(x, **case["decode"])
    return dec, out


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_sim_matches_reference_golden(case, sim_library):  # noqa: F811
    dec, out = run_case(case)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], case["expected"],
                tol=1e-9, what=case["name"])
    lm = dec._language_model
    for o, e in zip(out, case["expected"]):
        if e["state"] is None:
            assert o.last_lm_state is None
        else:
            st = o.last_lm_state.state
            assert [lm._kenlm_model.word(i) for i in st.words] == e["state"]["words"]
            assert [float(np.float32(b)) for b in st.backoff] == e["state"]["backoff"]
