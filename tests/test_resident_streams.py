"""Device-resident streams (ctcdec_stream_*, the lazy lists of partial_decode_beams): CPU checks through the simulator build;
the `-m gpu` twins at the bottom run the same scenarios on the HIP build. Reference: get_starting_state /
partial_decode_beams, decoder.py:669-728, and its tests (tests/test_decoder.py:515-698)."""
import numpy as np
import pytest

import synth
from oracle.ctc_oracle import build_oracle
from pyctcdecode_amd.alphabet import Alphabet
from pyctcdecode_amd.language_model import HotwordScorer
from tests.golden_util import LM_DIR, check_beams
from tests.sim_util import sim_library  # noqa: F401

LM = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
BPE = synth.make_bpe_vocab(LM.words, size=255)


def _oracle_chunks(orc, x, cuts, **kw):
    st = orc.get_starting_state()
    out = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        with np.errstate(all="ignore"):
            out = orc.partial_decode_beams(x[a:b].astype(np.float64), st, a, is_end=(b == cuts[-1]), **kw)
    return out


def _lm_beams(beams):
    return [(b.text + "|" + b.partial_word, [(str(k), f) for k, f in enumerate(b.text_frames)] + [("p", b.partial_frames)],
             b.logit_score, b.lm_score) for b in beams]


def _oracle_beams(ob):
    return [{"text": o.text + "|" + o.partial, "frames": [[str(k), int(f[0]), int(f[1])] for k, f in enumerate(o.tframes)]
             + [["p", int(o.pframes[0]), int(o.pframes[1])]], "logit": o.logit, "lm": o.lm} for o in ob]


def _scenario_unread_chunks(build, to_input, beam_width, labels, is_bpe):
    """The reference's usage pattern -- beams = partial_decode_beams(chunk, ..., beams, ...) -- without ever looking at the
    intermediate lists: nothing is materialised until the end, and the end equals the oracle's chunked decode."""
    dec = build(labels, LM.path)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    x = synth.d_words(3, 5, 180, labels, is_bpe, LM.words, LM.sentences, len(labels), boost=6.0).astype(np.float64)
    cuts = [0, 50, 51, 120, 180]
    kw = {"beam_width": beam_width, "prune_history": True}
    beams, c1, c2 = dec.get_starting_state()
    seen = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        beams = dec.partial_decode_beams(to_input(x[a:b]), c1, c2, beams, a, is_end=(b == 180), **kw)
        seen.append(beams)
    from pyctcdecode_amd.decoder import _ResidentBeams

    assert all(isinstance(s, _ResidentBeams) and not s._filled for s in seen[:-1])  # never looked at, never built
    assert not isinstance(seen[-1], _ResidentBeams)
    check_beams(_lm_beams(beams), _oracle_beams(_oracle_chunks(orc, x, cuts, **kw)), what="unread chunks")
    with pytest.raises(RuntimeError):  # the stream has moved on
        len(seen[0])
    return dec


def _scenario_reads_edits_and_hotwords(build, to_input):
    """Lists that are read stay valid and can still be handed back; edited lists, lists of another stream and a changed
    hot-word set go through the host import -- every route equals the oracle."""
    labels = synth.LIBRI_LABELS
    dec = build(labels, LM.path)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    x = synth.d_words(2, 9, 90, labels, False, LM.words, LM.sentences, 28, boost=6.0).astype(np.float64)
    hot = LM.hotwords(4, 1)
    scorer = HotwordScorer.build_scorer(hot, weight=8.0)
    cuts = [0, 30, 60, 90]
    # (a) read every chunk, hand the same list back
    beams, c1, c2 = dec.get_starting_state()
    st = orc.get_starting_state()
    for a, b in zip(cuts[:-1], cuts[1:]):
        beams = dec.partial_decode_beams(to_input(x[a:b]), c1, c2, beams, a, is_end=(b == 90), hotword_scorer=scorer)
        with np.errstate(all="ignore"):
            ob = orc.partial_decode_beams(x[a:b], st, a, is_end=(b == 90), hotwords=hot, hotword_weight=8.0)
        check_beams(_lm_beams(beams), _oracle_beams(ob), what="read %d:%d" % (a, b))
    # (b) keep only the best 7 beams after the first chunk (an edited list: host import), then go on
    beams, c1, c2 = dec.get_starting_state()
    st = orc.get_starting_state()
    beams = dec.partial_decode_beams(to_input(x[:30]), c1, c2, beams, 0)
    with np.errstate(all="ignore"):
        orc.partial_decode_beams(x[:30], st, 0)
    beams = beams[:7]
    st.beams = st.beams[:7]
    beams = dec.partial_decode_beams(to_input(x[30:60]), c1, c2, beams, 30)
    with np.errstate(all="ignore"):
        ob = orc.partial_decode_beams(x[30:60], st, 30)
    check_beams(_lm_beams(beams), _oracle_beams(ob), what="edited list")
    # (c) ... and resident again from there, with force_next_word in the middle of the stream
    beams = dec.partial_decode_beams(to_input(x[60:75]), c1, c2, beams, 60, force_next_word=True)
    with np.errstate(all="ignore"):
        ob = orc.partial_decode_beams(x[60:75], st, 60, force_next_word=True)
    check_beams(_lm_beams(beams), _oracle_beams(ob), what="force_next_word")
    beams = dec.partial_decode_beams(to_input(x[75:]), c1, c2, beams, 75, is_end=True)
    with np.errstate(all="ignore"):
        ob = orc.partial_decode_beams(x[75:], st, 75, is_end=True)
    check_beams(_lm_beams(beams), _oracle_beams(ob), what="after force_next_word")


def _scenario_many_streams_long(build, to_input, n_streams, n_chunks, chunk, beam_width):
    """Several streams in one launch per chunk, enough chunks for the emission arena to grow; compared with the unchunked
    decode of the same utterances (and stream 0 with the oracle)."""
    dec = build(BPE, LM.path)
    T = n_chunks * chunk
    xs = [synth.d_words(5, u, T, BPE, True, LM.words, LM.sentences, len(BPE), boost=6.0).astype(np.float64) for u in range(n_streams)]
    states = [dec.get_starting_state() for _ in range(n_streams)]
    beams = [s[0] for s in states]
    for k in range(n_chunks):
        beams = dec.partial_decode_beams_batch([to_input(x[k * chunk:(k + 1) * chunk]) for x in xs], [s[1] for s in states],
                                               [s[2] for s in states], beams, [k * chunk] * n_streams, beam_width=beam_width,
                                               is_end=(k == n_chunks - 1))
    whole = dec.decode_beams_batch(None, xs, beam_width=beam_width)
    for u in range(n_streams):
        assert [b.text for b in beams[u]] == [w.text for w in whole[u]]
        assert [b.text_frames for b in beams[u]] == [[f[1] for f in w.text_frames] for w in whole[u]]
        assert all(abs(b.lm_score - w.lm_score) < 1e-9 for b, w in zip(beams[u], whole[u]))
    alpha = Alphabet.build_alphabet(BPE)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    exp = _oracle_chunks(orc, xs[0], [k * chunk for k in range(n_chunks + 1)], beam_width=beam_width)
    check_beams(_lm_beams(beams[0]), _oracle_beams(exp), what="stream 0")


def _scenario_best_beam_per_chunk(build, to_input):
    """A caller that shows the transcript so far reads beams[0] after every chunk: that is a cheap read of the best beam
    alone (the list stays unread and the stream resident), it is the first beam of the full list, and the stream ends
    exactly where an unwatched one ends."""
    from pyctcdecode_amd.decoder import _ResidentBeams

    dec = build(BPE, LM.path)
    alpha = Alphabet.build_alphabet(BPE)
    orc = build_oracle(alpha.labels, alpha.is_bpe, LM.path, None)
    x = synth.d_words(4, 2, 160, BPE, True, LM.words, LM.sentences, len(BPE), boost=6.0).astype(np.float64)
    cuts = [0, 40, 80, 120, 160]
    beams, c1, c2 = dec.get_starting_state()
    st = orc.get_starting_state()
    for a, b in zip(cuts[:-1], cuts[1:]):
        beams = dec.partial_decode_beams(to_input(x[a:b]), c1, c2, beams, a, is_end=(b == 160), beam_width=50)
        with np.errstate(all="ignore"):
            ob = orc.partial_decode_beams(x[a:b], st, a, is_end=(b == 160), beam_width=50)
        best = beams[0]
        assert (best.text, best.partial_word) == (ob[0].text, ob[0].partial) and abs(best.lm_score - ob[0].lm) < 1e-9
        if b == 80:  # the full list after a peek: same first beam
            assert isinstance(beams, _ResidentBeams) and not beams._filled
            assert list(beams)[0] == best and len(beams) == len(ob)
        elif b < 160:
            assert isinstance(beams, _ResidentBeams) and not beams._filled
    check_beams(_lm_beams(beams), _oracle_beams(ob), what="watched stream")


def _build():
    from pyctcdecode_amd import build_ctcdecoder

    return build_ctcdecoder


@pytest.mark.parametrize("beam_width,labels,is_bpe", [(40, synth.LIBRI_LABELS, False), (100, BPE, True), (200, BPE, True)])
def test_unread_chunks_stay_on_the_device(beam_width, labels, is_bpe, sim_library, both_beam_kernels):  # noqa: F811
    _scenario_unread_chunks(_build(), lambda a: a, beam_width, labels, is_bpe)


def test_reads_edits_force_next_word_and_hotwords(sim_library, both_beam_kernels):  # noqa: F811
    _scenario_reads_edits_and_hotwords(_build(), lambda a: a)


def test_many_streams_and_a_growing_history(sim_library, both_beam_kernels):  # noqa: F811
    _scenario_many_streams_long(_build(), lambda a: a, n_streams=5, n_chunks=12, chunk=25, beam_width=30)


def test_best_beam_per_chunk_is_a_cheap_read(sim_library, both_beam_kernels):  # noqa: F811
    _scenario_best_beam_per_chunk(_build(), lambda a: a)


def test_lm_beams_built_in_c_equal_the_python_ones(sim_library, monkeypatch):  # noqa: F811
    """The LMBeam lists of a stream read are built in one C loop (csrc/pytexts.c: ctcdec_py_lm_beams); CTCDEC_PY_UNPACK=1 takes
    the Python loop it replaces: same beams field by field, same memo entries (cached_lm_scores) in the caller's dicts."""

    def run(py):
        if py:
            monkeypatch.setenv("CTCDEC_PY_UNPACK", "1")
        else:
            monkeypatch.delenv("CTCDEC_PY_UNPACK", raising=False)
        dec = _build()(BPE, LM.path)
        n = 5
        xs = [synth.d_words(5, u, 120, BPE, True, LM.words, LM.sentences, len(BPE), boost=4.0) for u in range(n)]
        states = [dec.get_starting_state() for _ in range(n)]
        beams = [s[0] for s in states]
        outs = []
        for k in range(3):
            beams = dec.partial_decode_beams_batch([x[k * 40:(k + 1) * 40] for x in xs], [s[1] for s in states], [s[2] for s in states],
                                                   beams, [k * 40] * n, beam_width=60, is_end=(k == 2))
            outs.append([[(type(b).__name__, b.text, b.next_word, b.partial_word, b.last_char, b.text_frames, b.partial_frames,
                           b.logit_score, b.lm_score) for b in bl] for bl in beams])
        memos = [sorted((k[0], v[0], v[1], bytes(v[2].state.to_c())) for k, v in s[1].items()) for s in states]
        return outs, memos

    assert run(False) == run(True)


def test_plain_lists_on_request(sim_library, monkeypatch):  # noqa: F811
    """CTCDEC_RESIDENT_STREAMS=0: every call returns ordinary, filled lists (and fills the caller's memo) like the reference."""
    from pyctcdecode_amd.decoder import _ResidentBeams

    monkeypatch.setenv("CTCDEC_RESIDENT_STREAMS", "0")
    dec = _build()(synth.LIBRI_LABELS, LM.path)
    x = synth.d_words(2, 1, 40, synth.LIBRI_LABELS, False, LM.words, LM.sentences, 28, boost=6.0)
    beams, c1, c2 = dec.get_starting_state()
    first = dec.partial_decode_beams(x[:20], c1, c2, beams, 0)
    assert not isinstance(first, _ResidentBeams) and len(first) > 0 and (first[0].text, False) in c1
    second = dec.partial_decode_beams(x[20:], c1, c2, first, 20, is_end=True)
    assert first[0].text is not None and second[0].text == dec.decode_beams(x)[0].text


def test_stream_seeded_with_the_lm_state_of_a_previous_segment(sim_library, both_beam_kernels):  # noqa: F811
    """A caller that starts a stream from get_starting_state() but puts the last_lm_state of an earlier segment into the
    memo's entry for the empty text (what decode_beams(lm_start_state=...) does, decoder.py:640-651) must be decoded from
    THAT state, not from the beginning of a sentence (round-3 advisor finding: the starting-state fast path ignored it)."""
    labels = synth.LIBRI_LABELS
    dec = _build()(labels, LM.path)
    x1 = synth.d_words(2, 3, 60, labels, False, LM.words, LM.sentences, 28, boost=6.0).astype(np.float64)
    x2 = synth.d_words(2, 4, 60, labels, False, LM.words, LM.sentences, 28, boost=6.0).astype(np.float64)
    state = dec.decode_beams(x1)[0].last_lm_state
    want = dec.decode_beams(x2, lm_start_state=state)
    cold = dec.decode_beams(x2)
    assert [w.lm_score for w in want] != [c.lm_score for c in cold]  # (the seed matters on this input)
    beams, c1, c2 = dec.get_starting_state()
    c1[("", False)] = (0.0, 0.0, state)
    got = dec.partial_decode_beams(x2, c1, c2, beams, 0, is_end=True)
    assert [g.text for g in got] == [w.text for w in want]
    assert all(abs(g.lm_score - w.lm_score) < 1e-9 and abs(g.logit_score - w.logit_score) < 1e-9 for g, w in zip(got, want))
    # ... and an untouched starting state still takes the resident fast path
    beams, c1, c2 = dec.get_starting_state()
    assert dec._memo_starts_at_default(c1)
    got = dec.partial_decode_beams(x2, c1, c2, beams, 0, is_end=True)
    assert [g.text for g in got] == [c.text for c in cold]


def test_lazy_lists_pickle_as_plain_lists_and_survive_a_refused_push(sim_library):  # noqa: F811
    import copy
    import pickle

    from pyctcdecode_amd.decoder import _ResidentBeams

    labels = synth.LIBRI_LABELS
    dec = _build()(labels, LM.path)
    x = synth.d_words(2, 6, 60, labels, False, LM.words, LM.sentences, 28, boost=6.0)
    beams, c1, c2 = dec.get_starting_state()
    lazy = dec.partial_decode_beams(x[:30], c1, c2, beams, 0)
    assert isinstance(lazy, _ResidentBeams) and not lazy._filled
    back = pickle.loads(pickle.dumps(lazy))
    assert type(back) is list and len(back) > 0 and back == list(lazy) and copy.copy(lazy) == back
    # a push the library refuses before anything runs leaves the previous chunk's (unread) lists readable
    lazy2 = dec.partial_decode_beams(x[30:45], c1, c2, lazy, 30)
    with pytest.raises(Exception):
        dec.partial_decode_beams(x[45:], c1, c2, lazy2, 45, beam_width=300)
    assert len(lazy2) > 0 and lazy2[0].text is not None


def test_multi_lm_streams_are_resident_too(sim_library):  # noqa: F811
    """Two language models (the workgroup kernel): the states of model 1.. ride along on the device."""
    from tests import test_multi_lm as M

    case = {"labels": BPE, "members": [{"lm": {"n_words": 300, "n_sent": 400, "order": 4, "seed": 2}},
                                       {"lm": {"n_words": 200, "n_sent": 300, "order": 3, "seed": 3},
                                        "build": {"alpha": 0.8, "beta": 1.0}}]}
    dec, _ = M.build_product_multi(case)
    orc = M.build_oracle_multi(case)
    x = synth.d_words(4, 3, 90, BPE, True, LM.words, LM.sentences, len(BPE), boost=6.0).astype(np.float64)
    cuts = [0, 31, 62, 90]
    beams, c1, c2 = dec.get_starting_state()
    for a, b in zip(cuts[:-1], cuts[1:]):
        beams = dec.partial_decode_beams(x[a:b], c1, c2, beams, a, is_end=(b == 90), beam_width=50)
    check_beams(_lm_beams(beams), _oracle_beams(_oracle_chunks(orc, x, cuts, beam_width=50)), what="multi")


# ---- the same scenarios on the HIP build ---------------------------------------------------------------------------------
def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("beam_width,labels,is_bpe", [(40, synth.LIBRI_LABELS, False), (100, BPE, True), (200, BPE, True)])
def test_hip_unread_chunks_stay_on_the_device(beam_width, labels, is_bpe, both_beam_kernels):
    _scenario_unread_chunks(_build(), _dev, beam_width, labels, is_bpe)


@pytest.mark.gpu
def test_hip_reads_edits_force_next_word_and_hotwords(both_beam_kernels):
    _scenario_reads_edits_and_hotwords(_build(), _dev)
    _scenario_reads_edits_and_hotwords(_build(), lambda a: a)  # host chunks


@pytest.mark.gpu
def test_hip_many_streams_and_a_growing_history(both_beam_kernels):
    _scenario_many_streams_long(_build(), _dev, n_streams=64, n_chunks=20, chunk=50, beam_width=200)


@pytest.mark.gpu
def test_hip_best_beam_per_chunk_is_a_cheap_read(both_beam_kernels):
    _scenario_best_beam_per_chunk(_build(), _dev)


@pytest.mark.gpu
def test_hip_bench_n_gt_1_path_rehearsed_on_one_gpu():
    """The N > 1 path of bench.py -- process group over RCCL (backend "nccl"), device binding by LOCAL_RANK, the one
    all_gather of texts per step -- at world size 1 (CTC_BENCH_FORCE_DIST=1), weak and strong scaling."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for scaling in ("weak", "strong"):
        env = dict(os.environ, CTC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1",
                   LOCAL_RANK="0")
        env.pop("CTCDEC_BEAM_KERNEL", None)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "192", "--frames", "200",
                              "--steps", "2", "--warmup", "1", "--scaling", scaling], env=env, capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 1 and line["scaling"] == scaling and line["value"] > 0  # n_gpus = what the process group reports
        assert "RCCL all_gather" in line["config"]["parallelism"] or line["n_gpus"] == 1
        # every rank's own step and stage times ride in the N > 1 line (an imbalance between the GPUs must be visible), and the
        # batch came from the bounded-memory chunked generator
        assert [r["rank"] for r in line["per_rank_stages_ms"]] == [0] and line["per_rank_stages_ms"][0]["beam"] > 0
        assert "generated in chunks" in out.stderr
    # `--gpus 2` on a box with one GPU must fail loudly, not fall back to one rank
    import torch

    if torch.cuda.device_count() == 1:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CTC_BENCH_FORCE_DIST")}
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "64", "--frames", "100",
                              "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--no-shard", "--no-peaky"],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode != 0
        assert "GPU(s) visible" in (out.stderr + out.stdout)
        # ... and a WORLD_SIZE that contradicts --gpus is refused too
        env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "64", "--frames", "100"],
                             env=env2, capture_output=True, text=True, timeout=600)
        assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_more_carried_beams_than_the_wave_table_holds(sim_library, monkeypatch):  # noqa: F811
    """A chunk decoded with beam_width 128 hands on up to 128 beams; the next one asks for beam_width 100, whose one-wave
    table holds 100 records: the launch has to fall back to the workgroup kernel instead of overrunning the table
    (round-2 advisor finding), resident and host-imported alike."""
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "wave")
    labels = synth.LIBRI_LABELS
    dec = _build()(labels)
    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe)
    x = synth.d_flat(2, 77, 24, 29).astype(np.float64)  # flat rows: the beam is full after a few frames
    for edit in (False, True):
        beams, c1, c2 = dec.get_starting_state()
        st = orc.get_starting_state()
        beams = dec.partial_decode_beams(x[:12], c1, c2, beams, 0, beam_width=128, beam_prune_logp=-30.0)
        with np.errstate(all="ignore"):
            orc.partial_decode_beams(x[:12], st, 0, beam_width=128, beam_prune_logp=-30.0)
        assert len(st.beams) > 100
        if edit:
            beams = list(beams)  # a plain list: the host import path
        beams = dec.partial_decode_beams(x[12:], c1, c2, beams, 12, beam_width=100, beam_prune_logp=-30.0, is_end=True)
        with np.errstate(all="ignore"):
            ob = orc.partial_decode_beams(x[12:], st, 12, beam_width=100, beam_prune_logp=-30.0, is_end=True)
        check_beams(_lm_beams(beams), _oracle_beams(ob), what="carry 128 -> 100")


def test_lazy_text_frames_and_memo_entries_behave_like_the_plain_objects():
    """The text_frames of a streaming read are windows of the read's arrays until looked at (decoder._LazyFrames), the memo
    entries build their LM-state objects on demand (decoder._MemoEntry): every way of looking must see the plain list / tuple."""
    import copy
    import pickle

    from pyctcdecode_amd.decoder import _MemoEntry, _lazy_frames_factory

    ws, we = np.array([0, 7, 20], dtype=np.int32), np.array([6, 13, 25], dtype=np.int32)
    make = _lazy_frames_factory(ws, we)
    plain = [(0, 6), (7, 13)]
    assert make(0, 2) == plain and plain == make(0, 2) and make(0, 2) == make(0, 2) and make(0, 2) != make(1, 3)
    assert len(make(0, 3)) == 3 and make(1, 3)[0] == (7, 13) and list(make(2, 3)) == [(20, 25)] and make(1, 1) == []
    assert make(0, 2) + make(2, 3) == plain + [(20, 25)] and plain + make(2, 3) == plain + [(20, 25)]
    f = make(0, 2)
    f.append((30, 31))
    assert f == plain + [(30, 31)] and isinstance(f[0][0], int)
    assert pickle.loads(pickle.dumps(make(0, 2))) == plain and type(pickle.loads(pickle.dumps(make(0, 2)))) is list
    assert copy.deepcopy(make(0, 2)) == plain and repr(make(0, 2)) == repr(plain) and (7, 13) in make(0, 2)
    # an OutputBeam's frames: the words of the beam's text paired with their (start, end) when somebody looks
    from pyctcdecode_amd.decoder import _lazy_word_frames_factory

    wmake = _lazy_word_frames_factory(ws, we)
    assert wmake("bugs bunny", 0, 2) == [("bugs", (0, 6)), ("bunny", (7, 13))] and wmake("", 0, 0) == []
    assert [w for w, _ in wmake("a b c", 0, 3)] == ["a", "b", "c"] and wmake("x", 2, 3)[0] == ("x", (20, 25))
    # memo entries: the state object is made from the read's packed states on first use, once
    import ctypes as C

    from pyctcdecode_amd import _binding as B
    from pyctcdecode_amd.decoder import _LazyMemo, _StateStore
    from pyctcdecode_amd.language_model import KenlmState

    raw_states = (B.LmState * 3)()
    raw_states[1].length = 2
    store = _StateStore(C.string_at(C.addressof(raw_states), C.sizeof(raw_states)))
    e = _MemoEntry.make(-1.5, store, 1)
    assert len(e) == 3 and not store._made
    lm_hw, raw, st = e
    assert (lm_hw, raw) == (-1.5, -1.5) and isinstance(st, KenlmState) and st.state.to_c().length == 2
    assert e[2] is st and len(store._made) == 1 and e == (-1.5, -1.5, st) and e == _MemoEntry.make(-1.5, store, 1)
    assert pickle.loads(pickle.dumps(e))[:2] == (-1.5, -1.5) and type(pickle.loads(pickle.dumps(e))) is tuple

    # the cache get_starting_state() hands out: reads are noted in O(1) and become entries when somebody looks
    class _B:  # (what a returned beam is to the memo: its text)
        def __init__(self, text):
            self.text = text

    memo = _LazyMemo({("", False): (0.0, 0.0, "START")})
    memo._note([_B("a"), _B("a b"), _B("")], [-1.0, -2.0, -9.0], store)
    assert memo._pending and dict.__len__(memo) == 1  # nothing filed yet
    assert ("a b", False) in memo and not memo._pending and len(memo) == 3
    assert memo[("a", False)][0] == -1.0 and memo.get(("a b", False))[1] == -2.0
    assert memo[("", False)] == (0.0, 0.0, "START")  # (an entry that is already there is not replaced)
    memo._note([_B("c")], [-3.0], store)
    memo[("c", False)] = (1.0, 1.0, "MINE")  # a caller's own entry wins over a pending note
    assert memo[("c", False)] == (1.0, 1.0, "MINE") and sorted(k[0] for k in memo) == ["", "a", "a b", "c"]
    memo._note([_B("d")], [-4.0], store)
    plain = pickle.loads(pickle.dumps(memo))
    assert type(plain) is dict and ("d", False) in plain and len(plain) == 5
