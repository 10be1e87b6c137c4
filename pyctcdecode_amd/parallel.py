"""Multi-GPU shell: utterances are independent (the reference's batch path is a plain map,
decoder.py:856,944), so a batch is sharded one contiguous slice per rank with NO exchange during
decoding; the only collective is the final gather of the decoded texts (sizes first, payload
second) over torch.distributed (backend "nccl" == RCCL over xGMI on MI355X, "gloo" in CPU tests)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous, balanced slice [lo, hi) of rank `rank`."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_texts(texts: Sequence[str], device=None, group=None) -> Optional[List[str]]:
    """All ranks call this with their shard's texts (in shard order); every rank gets the full list in
    global order (all_gather keeps the call symmetric and needs no rank-0 special case)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return list(texts)
    world = dist.get_world_size(group)
    enc = [t.encode("utf-8") for t in texts]
    lens = np.array([len(e) for e in enc], dtype=np.int64)
    payload = np.frombuffer(b"".join(enc), dtype=np.uint8)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    meta = torch.tensor([len(enc), payload.size], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [m.cpu().numpy() for m in metas]
    max_n = int(max(m[0] for m in metas))
    max_b = int(max(m[1] for m in metas))
    lens_t = torch.zeros(max(max_n, 1), dtype=torch.int64, device=dev)
    if len(enc):
        lens_t[: len(enc)] = torch.from_numpy(lens).to(dev)
    pay_t = torch.zeros(max(max_b, 1), dtype=torch.uint8, device=dev)
    if payload.size:
        pay_t[: payload.size] = torch.from_numpy(payload.copy()).to(dev)
    all_lens = [torch.zeros_like(lens_t) for _ in range(world)]
    all_pay = [torch.zeros_like(pay_t) for _ in range(world)]
    dist.all_gather(all_lens, lens_t, group=group)
    dist.all_gather(all_pay, pay_t, group=group)
    out: List[str] = []
    for r in range(world):
        n, _ = int(metas[r][0]), int(metas[r][1])
        ls = all_lens[r].cpu().numpy()[:n]
        buf = all_pay[r].cpu().numpy().tobytes()
        pos = 0
        for ln in ls:
            out.append(buf[pos : pos + int(ln)].decode("utf-8"))
            pos += int(ln)
    return out


def decode_batch_sharded(decoder, logits_list, group=None, **kwargs) -> List[str]:
    """Each rank decodes its contiguous slice of `logits_list` on its own GPU; all ranks return the
    texts of the whole batch in input order."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return decoder.decode_batch(None, logits_list, **kwargs)
    lo, hi = shard_bounds(len(logits_list), dist.get_world_size(group), dist.get_rank(group))
    local = decoder.decode_batch(None, logits_list[lo:hi], **kwargs)
    return gather_texts(local, group=group)
