"""CPU: the workgroup kernel (csrc/beam_core.h) on as many cooperative fibers as the device launch has threads
(tests/sim/group_fibers.h) instead of the simulator's one sequential thread: every thread-count-dependent path runs as on
the device (chunks of at most one candidate per thread, tiled pair loops, the rows-by-threads split of the top-B selection),
workgroup barriers are rendezvous of all fibers, and between two barriers whole waves run one after the other -- downwards
or upwards -- so a phase that reads another wave's writes without a barrier in between computes with stale values (removing
the barrier behind the ranking sweep, for one, crashes these tests both ways). Against the oracle, continuous inputs =>
strict order."""
import numpy as np
import pytest

import synth
from tests.golden_util import check_beams, lm_path, load_cases
from tests.sim_util import sim_library  # noqa: F401
from tests.test_sim_vs_oracle import BPE, LM, _compare

CASES, INPUTS = load_cases()
PICK = ("toy_lm_autounigrams_prune60", "rand_hf_flat_nolm_8", "toy_nolm_16beams", "toy_history_prune")


@pytest.fixture(params=[(256, "down"), (256, "up"), (512, "down"), (64, "up")], ids=lambda p: "%d%s" % p)
def group_fibers(request, monkeypatch):
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "group")
    monkeypatch.setenv("CTCDEC_SIM_GROUP_THREADS", str(request.param[0]))
    monkeypatch.setenv("CTCDEC_SIM_GROUP_ORDER", request.param[1])
    return request.param


def test_stress_frames_of_thousands_of_candidates(group_fibers, sim_library):  # noqa: F811
    """29 labels x 100 beams: several chunks a frame, pool compactions, the large-set selection (512 threads: the device's
    1024-candidate chunks and 4096-slot merge table)."""
    x = synth.d_flat(2, 41, 30, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, None, x, dkw={"prune_history": True}, what="fibers flat %s" % (group_fibers,))
    x = synth.d_flat(3, 42, 24, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, LM.path, x, dkw={"beam_width": 128}, what="fibers flat lm %s" % (group_fibers,))


def test_words_bpe_lm_hotwords_and_history(group_fibers, sim_library):  # noqa: F811
    x = synth.d_words(4, 43, 50, BPE, True, LM.words, LM.sentences, len(BPE), boost=5.0).astype(np.float64)
    _compare(BPE, LM.path, x, dkw={"prune_history": True, "hotwords": LM.hotwords(4, 2)}, what="fibers bpe %s" % (group_fibers,))
    x = (synth.d_flat(5, 44, 20, len(BPE) + 1) * 1.5).astype(np.float64)
    _compare(BPE, LM.path, x, dkw={"beam_width": 16, "token_min_logp": -8.0, "beam_prune_logp": -30.0},
             what="fibers bpe flat %s" % (group_fibers,))


@pytest.mark.parametrize("name", PICK)
def test_reference_goldens(name, group_fibers, sim_library):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    case = [c for c in CASES if c["name"] == name][0]
    dec = build_ctcdecoder(case["labels"], lm_path(case["lm"]), case["unigrams"], **case["build"])
    out = dec.decode_beams(INPUTS[case["input"]], **case["decode"])
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], case["expected"], tol=1e-9,
                what="%s %s" % (name, group_fibers))


def test_streams_in_chunks(group_fibers, sim_library):  # noqa: F811
    """partial_decode_beams: carried beams, import / carry-out paths of the kernel, equal to the unchunked decode."""
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(BPE, LM.path)
    x = synth.d_words(4, 45, 60, BPE, True, LM.words, LM.sentences, len(BPE), boost=5.0).astype(np.float64)
    whole = dec.decode_beams(x, beam_width=40)
    beams, c1, c2 = dec.get_starting_state()
    for k in range(0, 60, 20):
        beams = dec.partial_decode_beams(x[k:k + 20], c1, c2, beams, k, beam_width=40, is_end=(k + 20 >= 60))
    assert [(b.text, list(b.text_frames)) for b in beams] == [(o.text, [f for _, f in o.text_frames]) for o in whole]
    assert all(abs(b.lm_score - o.lm_score) < 1e-9 for b, o in zip(beams, whole))
