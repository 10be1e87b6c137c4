"""Multi-GPU shell: utterances are independent (the reference's batch path is a plain map,
decoder.py:856,944), so a batch is sharded one contiguous slice per rank with NO exchange during
decoding; the only collective is the final gather of the results over torch.distributed (backend
"nccl" == RCCL over xGMI on MI355X, "gloo" in CPU tests).

One collective per step: every rank packs [count, payload bytes, item lengths, payload] into one byte
buffer of an agreed capacity and a single all_gather moves all of them; only when some rank's results do not
fit the agreed capacity (every rank sees that in the gathered headers) a second all_gather with the exact
maximum follows.  Without an agreed capacity the sizes go first (two collectives).
"""
from __future__ import annotations

import pickle
from typing import Any, List, Optional, Sequence

import numpy as np

_HDR = 16  # int64 item count, int64 payload bytes


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous, balanced-by-count slice [lo, hi) of rank `rank`."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_bounds_by_frames(frames: Sequence[int], world_size: int, rank: int):
    """Contiguous slice [lo, hi) of rank `rank`, balanced by the SUM OF FRAMES (ragged batches: the decode time of
    a shard is proportional to its frames). Slice r ends at the first utterance whose cumulative
    frame count reaches (r + 1) / world of the total; every rank computes the same bounds from the same list."""
    n = len(frames)
    if n == 0:
        return 0, 0
    cum = np.cumsum(np.asarray(frames, dtype=np.int64))
    total = int(cum[-1])
    if total == 0:
        return shard_bounds(n, world_size, rank)
    cuts = [0]
    for r in range(1, world_size):
        c = int(np.searchsorted(cum, total * r / world_size, side="left")) + 1
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


def _pack(items: Sequence[bytes]) -> np.ndarray:
    lens = np.array([len(b) for b in items], dtype=np.int32)
    payload = b"".join(items)
    buf = np.zeros(_HDR + 4 * len(items) + len(payload), dtype=np.uint8)
    buf[:_HDR].view(np.int64)[:] = (len(items), len(payload))
    buf[_HDR:_HDR + 4 * len(items)].view(np.int32)[:] = lens
    buf[_HDR + 4 * len(items):] = np.frombuffer(payload, dtype=np.uint8)
    return buf


def _unpack(buf: np.ndarray) -> List[bytes]:
    n, nbytes = (int(v) for v in buf[:_HDR].view(np.int64))
    lens = buf[_HDR:_HDR + 4 * n].view(np.int32)
    raw = buf[_HDR + 4 * n:_HDR + 4 * n + nbytes].tobytes()
    out, pos = [], 0
    for ln in lens:
        out.append(raw[pos:pos + int(ln)])
        pos += int(ln)
    return out


def _all_gather_bytes(mine: np.ndarray, cap: int, world: int, dev, group) -> np.ndarray:
    """One all_gather of `cap` bytes per rank; returns a [world, cap] uint8 array (host)."""
    import torch
    import torch.distributed as dist

    send = torch.zeros(cap, dtype=torch.uint8)
    k = min(cap, mine.size)
    send[:k] = torch.from_numpy(mine[:k])
    send = send.to(dev)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.cpu().numpy().reshape(world, cap)


def gather_blobs(items: Sequence[bytes], capacity: Optional[int] = None, device=None, group=None) -> List[bytes]:
    """All ranks call this with their shard's byte strings (in shard order); every rank gets all of them in rank
    order. `capacity`: bytes per rank every rank agrees on WITHOUT talking (same value everywhere) -- then this is
    one collective unless a rank overflows it."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return list(items)
    world = dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    mine = _pack(items)
    if capacity is None:  # sizes first: 16 bytes per rank, then the payload at the exact maximum
        heads = _all_gather_bytes(mine, _HDR, world, dev, group)
        need = max(_HDR + 4 * int(h[:8].view(np.int64)[0]) + int(h[8:16].view(np.int64)[0]) for h in heads)
        got = _all_gather_bytes(mine, need, world, dev, group)
    else:
        cap = max(int(capacity), _HDR)
        got = _all_gather_bytes(mine, cap, world, dev, group)
        need = max(_HDR + 4 * int(g[:8].view(np.int64)[0]) + int(g[8:16].view(np.int64)[0]) for g in got)
        if need > cap:  # (every rank sees the same headers, so every rank takes this branch together)
            got = _all_gather_bytes(mine, need, world, dev, group)
    out: List[bytes] = []
    for r in range(world):
        out.extend(_unpack(got[r]))
    return out


def text_capacity(n_items: int, frames: int) -> int:
    """Bytes per rank that hold the texts of `n_items` utterances of about `frames` frames in all ordinary cases
    (a decoded text is far shorter than its frame count; an overflow only costs the second collective)."""
    return _HDR + n_items * (4 + 64 + max(0, int(frames)) // 2)


def gather_texts(texts: Sequence[str], device=None, group=None, capacity: Optional[int] = None) -> List[str]:
    """Every rank gets the texts of all ranks, in rank order (= global order for contiguous shards)."""
    return [b.decode("utf-8") for b in gather_blobs([t.encode("utf-8") for t in texts], capacity, device, group)]


def gather_objects(objs: Sequence[Any], device=None, group=None, capacity: Optional[int] = None) -> List[Any]:
    """The same for arbitrary picklable results (decode_beams_batch: lists of OutputBeam)."""
    return [pickle.loads(b) for b in gather_blobs([pickle.dumps(o, protocol=4) for o in objs], capacity, device, group)]


def _frames_of(logits_list) -> List[int]:
    return [int(x.shape[0]) for x in logits_list]


def decode_batch_sharded(decoder, logits_list, group=None, **kwargs) -> List[str]:
    """Each rank decodes its contiguous slice of `logits_list` (balanced by frames) on its own GPU; all ranks return
    the texts of the whole batch in input order. One collective."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return decoder.decode_batch(None, logits_list, **kwargs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    frames = _frames_of(logits_list)
    spans = [shard_bounds_by_frames(frames, world, r) for r in range(world)]
    lo, hi = spans[rank]
    local = decoder.decode_batch(None, logits_list[lo:hi], **kwargs)
    cap = max(text_capacity(b - a, 0) + sum(frames[a:b]) // 2 for a, b in spans)  # the same on every rank
    return gather_texts(local, group=group, capacity=cap)


def decode_beams_batch_sharded(decoder, logits_list, group=None, **kwargs) -> List[List[Any]]:
    """decode_beams_batch over the ranks (decoder.py:801-857): every rank returns the beams of the whole batch in
    input order (beams carry last_lm_state=None, as in the reference's pool path)."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return decoder.decode_beams_batch(None, logits_list, **kwargs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds_by_frames(_frames_of(logits_list), world, rank)
    local = decoder.decode_beams_batch(None, logits_list[lo:hi], **kwargs)
    return gather_objects(local, group=group)


# ---------------------------------------------------------------------------------------------------------------------
# DevicePool: what `pool` is to the reference's decode_batch(pool, ...) -- ONE caller process, several devices
# ---------------------------------------------------------------------------------------------------------------------
def _device_worker(conn, device: int, decoder_dir: str, library: Optional[str]) -> None:
    """One worker process = one GPU (the native library binds one device per process): loads the saved decoder on its device and
    serves (method, logits, kwargs) requests until it is told to stop."""
    import os
    import traceback

    os.environ["CTCDEC_DEVICE"] = str(device)
    try:
        from pyctcdecode_amd import _binding as B

        if library:  # (tests: the CPU simulator build of the device code)
            B._LIB = B.Library(library)
        from pyctcdecode_amd.decoder import BeamSearchDecoderCTC

        dec = BeamSearchDecoderCTC.load_from_dir(decoder_dir)
        conn.send(("ready", device))
    except Exception:  # the parent raises what went wrong
        conn.send(("error", traceback.format_exc()))
        return
    while True:
        msg = conn.recv()
        if msg is None:
            return
        method, logits, kwargs = msg
        try:
            if method == "decode_batch":
                conn.send(("ok", dec.decode_batch(None, logits, **kwargs)))
            else:
                beams = dec.decode_beams_batch(None, logits, **kwargs)
                conn.send(("ok", [[(b.text, list(b.text_frames), b.logit_score, b.lm_score) for b in bs] for bs in beams]))
        except Exception as e:  # noqa: BLE001 (sent back as the exception the direct call would have raised)
            conn.send(("raise", e if isinstance(e, (ValueError, NotImplementedError, TypeError)) else RuntimeError(traceback.format_exc())))


class DevicePool:
    """`pool` for `decode_batch(pool, logits_list, ...)` / `decode_beams_batch(pool, ...)` on a node with several GPUs
    (decoder.py:895-945, 801-857: the reference scales inside ONE process by handing a multiprocessing pool in). The native
    library binds one device per process, so a DevicePool is one spawned worker process per device, each holding a replica of
    the decoder (saved with save_to_dir, loaded in the worker); a call shards the batch over the workers by frames (contiguous
    slices: results come back in input order), each worker decodes its slice in one launch on its own GPU, and the parent
    collects the results -- no exchange between devices during the decode. The matrices travel to the workers as host arrays
    through pipes: this is the convenience form for callers that hold numpy logits; a pipeline that already has its logits on
    the GPUs runs one process per GPU and `decode_batch_sharded` (above).

        pool = DevicePool(decoder, devices=[0, 1, 2, 3])
        texts = decoder.decode_batch(pool, logits_list)
        pool.close()
    """

    def __init__(self, decoder, devices: Optional[Sequence[int]] = None, library: Optional[str] = None, start_timeout: float = 300.0):
        import multiprocessing as mp
        import tempfile

        if devices is None:
            import torch

            devices = list(range(max(1, torch.cuda.device_count())))
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("a DevicePool needs at least one device")
        self._dir = tempfile.TemporaryDirectory(prefix="ctcdec_pool_")
        decoder.save_to_dir(self._dir.name)
        ctx = mp.get_context("spawn")  # (never fork a process that holds a HIP context)
        self._workers = []
        for d in self.devices:
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_device_worker, args=(child, d, self._dir.name, library), daemon=True)
            p.start()
            child.close()
            self._workers.append((p, parent))
        for p, conn in self._workers:
            if not conn.poll(start_timeout):
                self.close()
                raise RuntimeError("a DevicePool worker did not come up within %.0f s" % start_timeout)
            kind, what = conn.recv()
            if kind != "ready":
                self.close()
                raise RuntimeError("a DevicePool worker failed to start:\n%s" % what)

    def _map(self, method: str, logits_list, kwargs):
        logits_list = [np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x) for x in logits_list]
        n, world = len(logits_list), len(self._workers)
        spans = [shard_bounds_by_frames(_frames_of(logits_list), world, r) for r in range(world)]
        busy = []
        for (lo, hi), (_p, conn) in zip(spans, self._workers):
            if hi > lo:
                conn.send((method, logits_list[lo:hi], kwargs))
                busy.append(conn)
        out: List[Any] = []
        err = None
        for conn in busy:  # (every answer is collected before anything is raised: the workers stay in step)
            kind, what = conn.recv()
            if kind == "ok":
                out.extend(what)
            elif err is None:
                err = what
        if err is not None:
            raise err
        assert len(out) == n
        return out

    def decode_batch(self, logits_list, **kwargs) -> List[str]:
        return self._map("decode_batch", logits_list, kwargs)

    def decode_beams_batch(self, logits_list, **kwargs):
        from pyctcdecode_amd.decoder import OutputBeam

        return [[OutputBeam(t, None, f, lg, lm) for t, f, lg, lm in beams] for beams in self._map("decode_beams_batch", logits_list, kwargs)]

    def close(self) -> None:
        for p, conn in getattr(self, "_workers", []):
            try:
                conn.send(None)
            except (OSError, BrokenPipeError):
                pass
        for p, conn in getattr(self, "_workers", []):
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
            conn.close()
        self._workers = []
        if getattr(self, "_dir", None) is not None:
            self._dir.cleanup()
            self._dir = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
