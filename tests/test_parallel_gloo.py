"""CPU: the N>1 path (utterance sharding + the one text gather) on world_size=2 with gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyctcdecode_amd.parallel import decode_batch_sharded, gather_texts, shard_bounds


class _FakeDecoder:
    """Stands in for the GPU decoder: the text of an utterance is a function of its content only."""

    def decode_batch(self, pool, logits_list, **kw):
        return ["utt-%d-é%s" % (int(x[0]), "x" * int(x[0] % 5)) for x in logits_list]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = [[i] for i in range(n_items)]
        texts = decode_batch_sharded(_FakeDecoder(), items)
        lo, hi = shard_bounds(n_items, world, rank)
        mine = gather_texts(["r%d-%d" % (rank, k) for k in range(lo, hi)])
        with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
            f.write("\n".join(texts) + "\n--\n" + "\n".join(mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 2, 1])
def test_sharded_decode_world2(tmp_path, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    expect = _FakeDecoder().decode_batch(None, [[i] for i in range(n_items)])
    for r in range(world):
        a, b = open(tmp_path / ("r%d.txt" % r)).read().split("\n--\n")
        assert a.split("\n") == expect
        got = [s for s in b.split("\n") if s]
        assert got == ["r%d-%d" % (rr, k) for rr in range(world) for k in range(*shard_bounds(n_items, world, rr))]


def test_shard_bounds_cover_everything():
    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
