"""Time-sliced ingest of host batches (csrc/api.cpp: decode_host_sliced; the reference is called with numpy matrices,
decoder.py:730-775). Large host batches go to the device in time slices through the device-resident stream machinery, the
copy of slice k + 1 under the kernels of slice k. Whatever the slicing, the result must be the one-piece decode's: every
beam, every frame, every score bit for bit -- and inputs whose probability sniff (decoder.py:760, a property of the WHOLE
utterance) is within reach of any slice must fall back to the one-piece path by themselves.
CTCDEC_HOST_SLICES=n forces n slices on any size (what these tests do), =0 switches the path off."""
import numpy as np
import pytest

import synth
from tests.golden_util import LM_DIR
from tests.sim_util import sim_library  # noqa: F401

LM = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)


def _key(beams):
    return [(b.text, list(b.text_frames), b.logit_score, b.lm_score) for b in beams]


def _both(monkeypatch, fn, slices):
    monkeypatch.setenv("CTCDEC_HOST_SLICES", "0")
    whole = fn()
    monkeypatch.setenv("CTCDEC_HOST_SLICES", str(slices))
    return whole, fn()


def _check_ragged(build_ctcdecoder, monkeypatch, to_dtype):
    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    lens = [61, 7, 0, 33, 90, 2, 90, 45]
    xs = [to_dtype(synth.d_words(2, u, t, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=4.0)) if t
          else np.zeros((0, 29), dtype=to_dtype(np.zeros(1)).dtype) for u, t in enumerate(lens)]
    hot = LM.hotwords(4, 1)
    for slices in (2, 3, 7):
        whole, sliced = _both(monkeypatch, lambda: dec.decode_beams_batch(None, xs, beam_width=24, hotwords=hot, prune_history=True), slices)
        assert [_key(b) for b in whole] == [_key(b) for b in sliced], slices
        assert any(len(b) > 2 for b in whole)
        tw, ts = _both(monkeypatch, lambda: dec.decode_batch(None, xs, beam_width=24, hotwords=hot), slices)
        assert tw == ts and tw == [b[0].text if b else "" for b in dec.decode_beams_batch(None, xs, beam_width=24, hotwords=hot, prune_history=True)]
    # one utterance through decode_beams (its last_lm_state comes back too) and decode()
    x = xs[4]
    whole, sliced = _both(monkeypatch, lambda: dec.decode_beams(x, beam_width=24), 4)
    assert _key(whole) == _key(sliced)
    assert [b.last_lm_state.state for b in whole] == [b.last_lm_state.state for b in sliced]
    # ... from a start state
    st = whole[0].last_lm_state
    whole, sliced = _both(monkeypatch, lambda: dec.decode_beams(xs[3], beam_width=24, lm_start_state=st), 3)
    assert _key(whole) == _key(sliced)


def _check_contiguous_3d_and_probabilities(build_ctcdecoder, monkeypatch):
    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    batch = np.stack([synth.d_words(2, 10 + u, 50, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=4.0) for u in range(5)])
    whole, sliced = _both(monkeypatch, lambda: dec.decode_beams_batch(None, batch, beam_width=16), 4)  # one [B, T, V] array: 2-D copies
    assert [_key(b) for b in whole] == [_key(b) for b in sliced]
    # probabilities: every slice is within reach of "mean row sum = 1" -> the call decodes in one piece by itself
    e = np.exp(batch.astype(np.float64) - batch.max(axis=2, keepdims=True))
    probs = e / e.sum(axis=2, keepdims=True)
    whole, sliced = _both(monkeypatch, lambda: dec.decode_beams_batch(None, probs, beam_width=16), 4)
    assert [_key(b) for b in whole] == [_key(b) for b in sliced]
    # rows that sum to ~1 in the first half and to something else in the second: the sniff is the whole utterance's
    mixed = np.concatenate([probs[0][:25], batch[0][25:].astype(np.float64)])
    whole, sliced = _both(monkeypatch, lambda: dec.decode_beams(mixed, beam_width=16), 2)
    assert _key(whole) == _key(sliced)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sim_sliced_ingest_equals_one_piece(dtype, sim_library, both_beam_kernels, monkeypatch):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    _check_ragged(build_ctcdecoder, monkeypatch, lambda a: np.asarray(a, dtype=dtype))


def test_sim_sliced_ingest_3d_batch_and_probability_fallback(sim_library, monkeypatch, capfd):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    monkeypatch.setenv("CTCDEC_SLICE_TRACE", "1")
    _check_contiguous_3d_and_probabilities(build_ctcdecoder, monkeypatch)
    trace = capfd.readouterr().err
    # the path was really taken (three sliced calls), and two of them found probability-like rows and decoded in one piece
    assert trace.count("time-sliced ingest: 5 utterances, 4 slices") == 2 and trace.count("time-sliced ingest: 1 utterances, 2 slices") == 1
    assert trace.count("decoding in one piece") == 2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.float16])
def test_hip_sliced_ingest_equals_one_piece(dtype, both_beam_kernels, monkeypatch):
    from pyctcdecode_amd import build_ctcdecoder

    _check_ragged(build_ctcdecoder, monkeypatch, lambda a: np.asarray(a, dtype=dtype))
    _check_contiguous_3d_and_probabilities(build_ctcdecoder, monkeypatch)


@pytest.mark.gpu
def test_hip_large_host_batch_takes_the_sliced_path_by_itself(monkeypatch):
    """A host batch above the size threshold (here: a non-contiguous float32 view of a padded array, 600 MB) is sliced without
    being asked to, and its texts equal the device-resident decode of the same values."""
    import torch

    import bench
    from pyctcdecode_amd import build_ctcdecoder

    monkeypatch.delenv("CTCDEC_HOST_SLICES", raising=False)
    lm, labels, hot = bench.build_assets(bench.cache_dir() if hasattr(bench, "cache_dir") else "bench_cache", 20000, 60000)
    dec = build_ctcdecoder(labels, lm.path)
    xs = bench.make_batch(lm, labels, 0, 160, 1000, 6.0, 8)  # 160 x 1000 x 1024 float32 = 655 MB
    host = [xs[u] for u in range(xs.shape[0])]
    texts = dec.decode_batch(None, host, hotwords=hot)
    dev = dec.decode_batch(None, torch.from_numpy(xs).cuda(), hotwords=hot)
    assert texts == dev and len(set(texts)) > 100
