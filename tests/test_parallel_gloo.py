"""CPU: the N>1 path (utterance sharding + the one result gather) on world_size=2 with gloo."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from pyctcdecode_amd.parallel import (decode_batch_sharded, decode_beams_batch_sharded, gather_blobs, gather_texts,
                                      shard_bounds, shard_bounds_by_frames, text_capacity)


class _Item:
    """A stand-in logits matrix: only .shape[0] (frames) and its identity matter here."""

    def __init__(self, ident, frames):
        self.ident, self.shape = ident, (frames, 4)


class _FakeDecoder:
    """Stands in for the GPU decoder: the result of an utterance is a function of its content only."""

    def decode_batch(self, pool, logits_list, **kw):
        return ["utt-%d-é%s" % (x.ident, "x" * (x.ident % 5)) for x in logits_list]

    def decode_beams_batch(self, pool, logits_list, **kw):
        return [[("beam", x.ident, k, float(x.shape[0])) for k in range(1 + x.ident % 3)] for x in logits_list]


def _items(n):
    return [_Item(i, 10 + 37 * (i % 4)) for i in range(n)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_CALLS = []


def _worker(rank, world, port, n_items, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orig = dist.all_gather_into_tensor

    def counting(*a, **k):
        _CALLS.append(1)
        return orig(*a, **k)

    dist.all_gather_into_tensor = counting
    try:
        items = _items(n_items)
        texts = decode_batch_sharded(_FakeDecoder(), items)
        n_text_calls = len(_CALLS)
        beams = decode_beams_batch_sharded(_FakeDecoder(), items)
        lo, hi = shard_bounds(n_items, world, rank)
        mine = ["r%d-%d" % (rank, k) for k in range(lo, hi)]
        del _CALLS[:]
        agreed = gather_texts(mine, capacity=text_capacity(n_items, 100))       # fits: exactly one collective
        one = len(_CALLS)
        del _CALLS[:]
        tight = gather_texts(mine, capacity=16)                                 # overflows: a second one follows
        two = len(_CALLS)
        del _CALLS[:]
        sized = gather_texts(mine)                                              # no agreed capacity: sizes first
        three = len(_CALLS)
        blobs = gather_blobs([bytes([rank]) * (rank + 1)])
        with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
            f.write(repr({"texts": texts, "beams": beams, "agreed": agreed, "tight": tight, "sized": sized,
                          "calls": [n_text_calls, one, two, three], "blobs": blobs}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 2, 1])
def test_sharded_decode_world2(tmp_path, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    items = _items(n_items)
    expect = _FakeDecoder().decode_batch(None, items)
    expect_beams = _FakeDecoder().decode_beams_batch(None, items)
    mine = ["r%d-%d" % (rr, k) for rr in range(world) for k in range(*shard_bounds(n_items, world, rr))]
    for r in range(world):
        got = eval(open(tmp_path / ("r%d.txt" % r)).read())  # noqa: S307 (our own repr)
        assert got["texts"] == expect
        assert got["beams"] == expect_beams
        assert got["agreed"] == mine and got["tight"] == mine and got["sized"] == mine
        assert got["calls"] == [1, 1, 2, 2]  # decode_batch_sharded and an agreed capacity: ONE collective
        assert got["blobs"] == [b"\x00", b"\x01\x01"]


def _real_worker(rank, world, port, out_dir):
    """The REAL decoder (Python shell + C ABI + both beam kernels' source, on the CPU simulator backend) in every rank."""
    import synth
    from pyctcdecode_amd import _binding as B
    from pyctcdecode_amd import build_ctcdecoder
    from tests.golden_util import LM_DIR
    from tests.sim.build_sim import build

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B._LIB = B.Library(build())
        lm = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
        labels = synth.LIBRI_LABELS
        dec = build_ctcdecoder(labels, lm.path)
        # every rank holds the whole batch (ragged lengths) and decodes only its shard
        xs = [synth.d_words(2, u, 40 + 13 * (u % 4), labels, False, lm.words, lm.sentences, 28, boost=6.0) for u in range(7)]
        hot = lm.hotwords(3, 1)
        texts = decode_batch_sharded(dec, xs, beam_width=25, hotwords=hot)
        beams = decode_beams_batch_sharded(dec, xs, beam_width=25, hotwords=hot)
        balanced = decode_batch_sharded(dec, xs[::-1], beam_width=25, hotwords=hot)[::-1]  # (another frame balance)
        with open(os.path.join(out_dir, "real%d.txt" % rank), "w") as f:
            f.write(repr({"texts": texts, "balanced": balanced,
                          "beams": [[(b.text, b.text_frames, b.logit_score, b.lm_score) for b in bl] for bl in beams]}))
        if rank == 0:  # the unsharded result, from the same process
            whole = dec.decode_batch(None, xs, beam_width=25, hotwords=hot)
            whole_beams = dec.decode_beams_batch(None, xs, beam_width=25, hotwords=hot)
            with open(os.path.join(out_dir, "whole.txt"), "w") as f:
                f.write(repr({"texts": whole,
                              "beams": [[(b.text, b.text_frames, b.logit_score, b.lm_score) for b in bl] for bl in whole_beams]}))
    finally:
        dist.destroy_process_group()


def test_real_decoder_sharded_over_two_ranks_equals_the_unsharded_decode(tmp_path):
    """decode_batch_sharded / decode_beams_batch_sharded with the real decoder (simulator backend) in two gloo ranks:
    every rank ends up with the whole batch's results, equal to one unsharded decode_batch (decoder.py:856, 944)."""
    from tests.sim.build_sim import build

    build()  # (once, before the ranks race for it)
    world = 2
    mp.spawn(_real_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    whole = eval(open(tmp_path / "whole.txt").read())  # noqa: S307 (our own repr)
    assert len(whole["texts"]) == 7 and any(whole["texts"])
    for r in range(world):
        got = eval(open(tmp_path / ("real%d.txt" % r)).read())  # noqa: S307
        assert got["texts"] == whole["texts"]
        assert got["balanced"] == whole["texts"]
        assert got["beams"] == whole["beams"]


def test_shard_bounds_cover_everything():
    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_shard_bounds_by_frames_balance_ragged_batches():
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 64, 257):
        frames = [int(f) for f in rng.integers(0, 2000, size=n)]
        for w in (1, 2, 3, 8):
            spans = [shard_bounds_by_frames(frames, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            if n >= 8 * w:
                loads = [sum(frames[a:b]) for a, b in spans]
                assert max(loads) <= sum(frames) / w + max(frames)  # never worse than one utterance over the mean
    # equal lengths: the same as the count-balanced split up to one utterance
    spans = [shard_bounds_by_frames([100] * 10, 3, r) for r in range(3)]
    assert [b - a for a, b in spans] in ([4, 3, 3], [3, 4, 3], [3, 3, 4], [4, 4, 2])


def test_chunked_batch_generation_equals_the_whole_batch():
    """bench.py --gpus N > 1 generates each rank's batch chunk by chunk into one small shared buffer (bounded host memory on an
    8-GPU node): the chunks must be the very utterances bench.make_batch would have produced, bit for bit."""
    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = os.path.join(root, "bench_cache") if os.access(root, os.W_OK) else "/tmp/ctc_bench"
    lm, labels, hot = bench.build_assets(cache, 20000, 60000)
    whole = bench.make_batch(lm, labels, 37, 10, 40, 6.0, 1)
    cb = bench.ChunkedBatch(lm, labels, 40, 6.0, 2, chunk=4)
    try:
        got = []
        for c0 in range(0, 10, 4):
            n = min(4, 10 - c0)
            got.append(np.array(cb.fill(37 + c0, n)))  # (copied: the buffer is reused)
    finally:
        cb.close()
    assert np.array_equal(np.concatenate(got), whole)
