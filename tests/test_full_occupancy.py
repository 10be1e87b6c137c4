"""The wave kernel the way the bench runs it: ONE decode_beams_batch of >= 2048 utterances (sixteen waves per CU sharing the
LDS, per-utterance ColdRec / PoolPay lines and arena partitions in global memory), with the committed full-size and
real-posterior-like goldens of the unmodified reference (tests/golden/cases_full.json.gz, cases_peaky.json.gz: T=1000,
V=1024, 4-gram + hot words, EVERY beam with its frames) embedded at scattered places of the batch -- first, last, the
boundaries between CUs' shares -- among ragged filler utterances of the same generator (200-600 frames).

A call has one set of decode arguments and one dtype, so the goldens are grouped by (arguments, dtype) and each group gets a
full batch of its own.  Per batch:
  * the launcher picked the wave kernel by itself (no CTCDEC_BEAM_KERNEL) and reports it;
  * every beam of every embedded golden goes through check_beams against the reference's committed output (decoder.py:801-857);
  * the WHOLE batch -- all beams, both scores bit for bit, word frames -- equals the same batch under CTCDEC_BEAM_KERNEL=group
    (the workgroup kernel, which every other parity test of the suite pins);
  * 32 sampled places (the goldens' among them) equal a decode_beams() call of that utterance alone.
The CPU twin runs the same bookkeeping on the simulator with a handful of short fillers."""
import json
import os

import numpy as np
import pytest

import bench
import synth
from tests.sim_util import sim_library  # noqa: F401
from tests.test_golden_full import CASES, _check, _input, assets  # noqa: F401

N_UTTS = 2048 + 96  # > 2 x 256 CUs x the workgroup kernel's residency: the wave kernel's territory, 8+ waves per CU at once
N_DISTINCT = 536    # distinct fillers (each reused with four different lengths)


def _groups():
    """goldens of the bench vocabulary grouped by what one call can share"""
    out = {}
    for c in CASES:
        if c["kind"] == "config2":
            continue  # (29 labels, no LM: another decoder; its frames have thousands of candidates -- the workgroup kernel's case)
        out.setdefault((json.dumps(c["decode"], sort_keys=True), c["dtype"]), []).append(c)
    return out


def _places(n_gold, n_utts, cus=256):
    """where the goldens go: first, last, and the seams between consecutive CUs' shares of the launch (the hardware hands
    workgroups out in order: with n_utts / cus waves per CU these indices sit at the start / end of a CU's share)"""
    per_cu = max(1, n_utts // cus)
    want = [0, n_utts - 1, per_cu - 1, per_cu, n_utts // 2, n_utts // 2 + 1, 7 * per_cu + 3, n_utts - per_cu]
    seen, places = set(), []
    for p in want:
        p = min(max(p, 0), n_utts - 1)
        while p in seen:
            p = (p + 1) % n_utts
        seen.add(p)
        places.append(p)
    return places[:n_gold]


def _as_tuples(beams):
    return [(b.text, tuple((w, (int(s), int(e))) for w, (s, e) in b.text_frames), b.logit_score, b.lm_score) for b in beams]


def _run_group(dec, key, cases, assets_, fillers, n_utts, to_input, monkeypatch, n_sample=32):
    lm, labels, hot = assets_
    kw = dict(json.loads(key[0]))
    kw["hotwords"] = hot if kw.get("hotwords") == "bench" else None
    golden_inputs = []
    for c in cases:
        _, _, x, kw_c = _input(c, assets_)
        assert {k: v for k, v in kw_c.items() if k != "hotwords"} == {k: v for k, v in kw.items() if k != "hotwords"}
        golden_inputs.append(to_input(x))
    places = _places(len(cases), n_utts)
    rng = np.random.default_rng(17)
    max_t = fillers[0].shape[0]
    batch = []
    for u in range(n_utts):
        t = int(rng.integers(max_t // 3, max_t + 1))  # ragged: 200-600 frames on the device
        batch.append(fillers[u % len(fillers)][:t])
    for p, x in zip(places, golden_inputs):
        batch[p] = x
    monkeypatch.delenv("CTCDEC_BEAM_KERNEL", raising=False)
    out = dec.decode_beams_batch(None, batch, **kw)
    picked = dec.last_beam_kernel
    assert len(out) == n_utts
    tol = 1e-9 if key[1] == "float64" else 1e-4
    for p, c in zip(places, cases):
        _check(c, [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in out[p]], tol)
    wave = [_as_tuples(b) for b in out]
    del out
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "group")
    out_g = dec.decode_beams_batch(None, batch, **kw)
    assert dec.last_beam_kernel == 2
    group = [_as_tuples(b) for b in out_g]
    del out_g
    differ = [u for u in range(n_utts) if wave[u] != group[u]]
    assert not differ, "wave and workgroup kernels disagree at %d places, first %d: %r vs %r" % (
        len(differ), differ[0], wave[differ[0]][:2], group[differ[0]][:2])
    monkeypatch.delenv("CTCDEC_BEAM_KERNEL", raising=False)
    sample = list(places) + [int(i) for i in rng.choice(n_utts, size=max(0, n_sample - len(places)), replace=False)]
    for u in sample:
        alone = _as_tuples(dec.decode_beams(batch[u], **kw))
        assert alone == wave[u], "utterance %d alone differs from its place in the batch" % u
    return picked, sum(len(b) for b in wave)


@pytest.fixture(scope="module")
def filler_host(assets):  # noqa: F811
    lm, labels, hot = assets
    return np.stack([synth.d_words(bench.CONFIG_ID, 50_000 + u, 600, labels, True, lm.words, lm.sentences, len(labels), boost=6.0)
                     for u in range(N_DISTINCT)])


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(_groups()), ids=lambda k: "%s-%s" % (k[1], "".join(ch for ch in k[0] if ch.isalnum())[:40]))
def test_hip_wave_kernel_at_bench_occupancy(key, assets, filler_host, monkeypatch):  # noqa: F811
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    lm, labels, hot = assets
    cases = _groups()[key]
    tdt = torch.float64 if key[1] == "float64" else torch.float32
    dev = torch.from_numpy(filler_host).cuda().to(tdt)  # [N_DISTINCT, 600, V]: the ragged fillers are views of it
    fillers = [dev[i] for i in range(N_DISTINCT)]
    dec = build_ctcdecoder(labels, lm.path)
    picked, n_beams = _run_group(dec, key, cases, assets, fillers, N_UTTS, lambda x: torch.from_numpy(x).cuda(), monkeypatch)
    assert picked == 1, "the launcher did not pick the wave kernel for %d utterances" % N_UTTS
    print("%s: %d utterances in one call, %d goldens embedded, %d beams equal under both kernels" % (
        key[1], N_UTTS, len(cases), n_beams))


def test_sim_batch_bookkeeping_of_the_occupancy_test(assets, sim_library, monkeypatch):  # noqa: F811
    """The same harness on the simulator (24 utterances, short fillers, one group): places, grouping, the three comparisons."""
    from pyctcdecode_amd import build_ctcdecoder

    lm, labels, hot = assets
    key = sorted(k for k, v in _groups().items() if k[1] == "float64" and any(c["kind"] == "peaky" for c in v))[0]
    cases = [c for c in _groups()[key] if c["kind"] == "peaky"][:1]
    fillers = [synth.d_words(bench.CONFIG_ID, 50_000 + u, 60, labels, True, lm.words, lm.sentences, len(labels),
                             boost=6.0).astype(np.float64) for u in range(6)]
    dec = build_ctcdecoder(labels, lm.path)
    picked, n_beams = _run_group(dec, key, cases, assets, fillers, 24, lambda x: x, monkeypatch, n_sample=4)
    assert picked == 1 and n_beams >= 24
    assert len(set(_places(8, 2144))) == 8 and _places(3, 24)[:2] == [0, 23]


@pytest.mark.gpu
@pytest.mark.parametrize("n,frames", [(4 * 256 + 70, 90), (5000, 24), (8192 + 40, 16)],
                         ids=["one_round", "more_than_one_round", "more_than_the_ranking_holds"])
def test_hip_placement_by_weight_keeps_every_result(assets, filler_host, monkeypatch, n, frames):  # noqa: F811
    """Round 6: a launch of more than one wave per SIMD weighs its utterances (survivors per frame), dispatches the heavy ones
    first in a snake over the SIMDs and runs them at a higher issue priority (utt_weigh / utt_place, backend_hip.hip). Which
    workgroup decodes an utterance and at which priority must not show in any beam: an equal-length batch (ragged ones keep their
    longest-first order) under the default, under CTCDEC_WAVE_PRIO=dyn (no weights, no placement), with the weights but without
    the placement, and dealt out lightest first."""
    import torch

    from pyctcdecode_amd import build_ctcdecoder

    lm, labels, hot = assets
    # (more than one wave per SIMD: the placement is on; 5 000: more utterances than wave slots; 8 232: more than utt_place ranks --
    #  weights only, the order stays)
    dev = torch.from_numpy(filler_host[:, :frames]).cuda()  # [N_DISTINCT, frames, V]
    batch = torch.stack([dev[(7 * u) % N_DISTINCT].roll(u % 5, 0) for u in range(n)])  # (distinct survivor counts per utterance)
    dec = build_ctcdecoder(labels, lm.path)
    monkeypatch.setenv("CTCDEC_BEAM_KERNEL", "wave")
    kw = dict(beam_width=100, hotwords=hot)
    kw_beams = dict(kw, prune_history=True)  # (what decode_batch decodes with: decoder.py:944 -> decode())
    runs = {}
    for name, env in (("default", {}), ("dyn", {"CTCDEC_WAVE_PRIO": "dyn"}), ("noplace", {"CTCDEC_NO_PLACE": "1"}),
                      ("lightest first", {"CTCDEC_PLACE_SNAKE": "2", "CTCDEC_WEIGH_GAIN": "8"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        runs[name] = [_as_tuples(b) for b in dec.decode_beams_batch(None, batch, **kw_beams)]
        assert dec.last_beam_kernel == 1
        for k in env:
            monkeypatch.delenv(k)
    texts = dec.decode_batch(None, batch, **kw)
    assert texts == [b[0][0] for b in runs["default"]]
    for name, got in runs.items():
        differ = [u for u in range(n) if got[u] != runs["default"][u]]
        assert not differ, "%s: %d utterances differ from the default launch, first %d" % (name, len(differ), differ[0])
