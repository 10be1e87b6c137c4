"""CPU: randomised differential test of the shell + beam_core.h (sequential sim backend) against
the oracle at sizes that exercise chunking (> 512 candidates per frame), pool pruning, 3-way
merges, BPE force_next_break and LM/hot-word fusion.  Continuous inputs => strict order."""
import numpy as np
import pytest

import synth
from oracle.ctc_oracle import build_oracle
from pyctcdecode_amd.alphabet import Alphabet
from tests.golden_util import LM_DIR, check_beams
from tests.sim_util import sim_library  # noqa: F401


def _expected(orc, x, kw):
    with np.errstate(all="ignore"):
        out = orc.decode_beams(x, **kw)
    return [{"text": o[0], "frames": [[w, int(a), int(b)] for w, (a, b) in o[2]], "logit": o[3], "lm": o[4]} for o in out]


def _compare(labels, arpa, x, build=None, dkw=None, unigrams=None, what=""):
    from pyctcdecode_amd import build_ctcdecoder

    build = build or {}
    dkw = dkw or {}
    dec = build_ctcdecoder(labels, arpa, unigrams, **build)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe, arpa, unigrams, **build)
    out = dec.decode_beams(x, **dkw)
    check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out], _expected(orc, x, dkw),
                tol=1e-9, what=what)
    return dec, orc


LM = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
WORDS = synth.make_words(300, seed=2)
BPE = synth.make_bpe_vocab(WORDS, size=255)


@pytest.mark.parametrize("seed", range(6))
def test_flat_libri_beam100_chunks(seed, sim_library):  # noqa: F811
    x = synth.d_flat(2, seed, 40, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, None, x, dkw={"prune_history": bool(seed % 2)}, what="flat%d" % seed)


@pytest.mark.parametrize("seed", range(6))
def test_flat_libri_lm(seed, sim_library):  # noqa: F811
    x = synth.d_flat(3, seed, 30, 29).astype(np.float64)
    _compare(synth.LIBRI_LABELS, LM.path, x, dkw={"prune_history": bool(seed % 2), "beam_width": 60},
             what="flatlm%d" % seed)


@pytest.mark.parametrize("seed", range(6))
def test_words_bpe_lm_hotwords(seed, sim_library):  # noqa: F811
    x = synth.d_words(4, seed, 50, BPE, True, LM.words, LM.sentences, len(BPE), boost=5.0).astype(np.float64)
    hot = LM.hotwords(4, 2) if seed % 2 else None
    _compare(BPE, LM.path, x, dkw={"prune_history": seed % 3 != 0, "hotwords": hot}, what="bpe%d" % seed)


@pytest.mark.parametrize("seed", range(4))
def test_flat_bpe_small_beam_large_tokens(seed, sim_library):  # noqa: F811
    x = (synth.d_flat(5, seed, 25, len(BPE) + 1) * 1.5).astype(np.float64)
    _compare(BPE, LM.path, x, dkw={"beam_width": 16, "token_min_logp": -8.0, "beam_prune_logp": -30.0},
             what="bpeflat%d" % seed)


def test_decode_and_batch_api(sim_library):  # noqa: F811
    from pyctcdecode_amd import build_ctcdecoder

    dec = build_ctcdecoder(synth.LIBRI_LABELS, LM.path)
    alpha = Alphabet.build_alphabet(synth.LIBRI_LABELS)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    xs = [synth.d_words(2, u, 20 + 7 * u, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=6.0)
          for u in range(5)] + [np.zeros((0, 29), dtype=np.float32)]
    texts = dec.decode_batch(None, xs)
    assert texts == [orc.decode(x.astype(np.float64)) for x in xs]
    assert dec.decode(xs[1]) == texts[1]
    beams = dec.decode_beams_batch(None, xs[:2], beam_width=20)
    assert [b[0].text for b in beams] == [orc.decode_beams(x.astype(np.float64), beam_width=20)[0][0] for x in xs[:2]]
    assert all(b.last_lm_state is None for bb in beams for b in bb)
    with pytest.raises(ValueError):
        dec.decode(np.zeros((3, 30)))
    with pytest.raises(ValueError):
        dec.decode(np.zeros((3,)))
