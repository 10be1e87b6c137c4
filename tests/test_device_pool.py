"""parallel.DevicePool: `pool` for decode_batch(pool, ...) on several GPUs from ONE caller process (the reference scales
inside one process by handing a multiprocessing pool in, decoder.py:895-945) -- one spawned worker per device, each with a
replica of the decoder, the batch sharded by frames, results in input order. On CPU the workers run the simulator build of the
device code; `-m gpu`: two workers on the one device of the box (two processes, two HIP contexts)."""
import numpy as np
import pytest

import synth
from tests.golden_util import LM_DIR
from tests.sim_util import sim_library  # noqa: F401


def _check(dec, pool, xs, hot):
    want = dec.decode_beams_batch(None, xs, beam_width=16, hotwords=hot)
    got = dec.decode_beams_batch(pool, xs, beam_width=16, hotwords=hot)
    key = lambda beams: [[(b.text, list(b.text_frames), b.logit_score, b.lm_score, b.last_lm_state) for b in bs] for bs in beams]  # noqa: E731
    assert key(got) == key(want) and any(len(b) > 1 for b in want)
    assert dec.decode_batch(pool, xs, beam_width=16, hotwords=hot) == dec.decode_batch(None, xs, beam_width=16, hotwords=hot)
    assert dec.decode_batch(pool, xs[:1]) == dec.decode_batch(None, xs[:1])  # fewer utterances than workers
    with pytest.raises(ValueError):  # the workers' errors are the direct call's
        dec.decode_batch(pool, [np.zeros((5, 3))])


def test_device_pool_on_two_simulator_workers(sim_library):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.language_model import load_unigram_set_from_arpa
    from pyctcdecode_amd.parallel import DevicePool
    from tests.sim.build_sim import build

    lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
    dec = build_ctcdecoder(synth.LIBRI_LABELS, lm.path, load_unigram_set_from_arpa(lm.path))
    xs = [synth.d_words(2, u, t, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=4.0) for u, t in enumerate([40, 9, 33, 21, 40])]
    with DevicePool(dec, devices=[0, 0], library=build()) as pool:
        _check(dec, pool, xs, lm.hotwords(3, 1))


@pytest.mark.gpu
def test_hip_device_pool_two_workers_on_one_gpu():
    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.language_model import load_unigram_set_from_arpa
    from pyctcdecode_amd.parallel import DevicePool

    lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
    dec = build_ctcdecoder(synth.LIBRI_LABELS, lm.path, load_unigram_set_from_arpa(lm.path))
    xs = [synth.d_words(2, u, t, synth.LIBRI_LABELS, False, lm.words, lm.sentences, 28, boost=4.0) for u, t in enumerate([60, 9, 33, 21, 80, 44])]
    with DevicePool(dec, devices=[0, 0]) as pool:
        _check(dec, pool, xs, lm.hotwords(3, 1))
