#!/usr/bin/env python
"""bench.py -- logit-frames/s of decode_batch on MI355X (BASELINE.json metric).

Workload ("step" = one decode_batch over one batch of synthetic logits already resident in HBM):
BASELINE config "Conformer-CTC BPE vocab V=1024, 4-gram LM + hotword boost, beam=100, batch=4096
[T=1000] sharded over 8 GPUs" => 512 utterances x T=1000 x V=1024 per GPU (weak scaling: each rank
decodes its own 512 utterances, texts are gathered over RCCL at the end of every step).
Inputs: D_words(boost 6.0) of SURVEY 8(d) (seeded, synthetic), fp32 logits, 20k-word synthetic
4-gram ARPA, 20 in-vocabulary + 5 OOV hot words.

Usage: python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import synth  # noqa: E402

CONFIG_ID = 4
V = 1024
T = 1000
BEAM = 100
HBM_PEAK = 8.0e12  # B/s (MI355X_MICROARCH.md: 8 TB/s spec)


def build_assets(cache_dir, n_words, n_sent):
    lm = synth.SynthLM(cache_dir, n_words, n_sent, order=4, seed=7,
                       max_ngrams={2: 200_000, 3: 400_000, 4: 400_000})
    labels = synth.make_bpe_vocab(lm.words, size=V - 1)
    hot = lm.hotwords(20, 5)
    return lm, labels, hot


def make_batch(lm, labels, first_utt, n_utts, frames, boost=6.0):
    return [synth.d_words(CONFIG_ID, first_utt + u, frames, labels, True, lm.words, lm.sentences, len(labels),
                          boost=boost) for u in range(n_utts)]


def cpu_baseline(lm, labels, hot, xs, cores):
    """The oracle's decode_batch (same algorithm and cost structure as the reference's pure-Python
    decode_batch) on a fork pool over the host cores, on a bounded sample of the same workload."""
    import multiprocessing as mp

    from oracle.ctc_oracle import build_oracle
    from pyctcdecode_amd.alphabet import Alphabet

    alpha = Alphabet.build_alphabet(labels)
    orc = build_oracle(alpha.labels, alpha.is_bpe, lm.path, None)
    xs64 = [x.astype(np.float64) for x in xs]
    frames = sum(x.shape[0] for x in xs64)
    with mp.get_context("fork").Pool(cores) as pool:  # created after the decoder (README.md:80-85)
        t0 = time.perf_counter()
        texts = orc.decode_batch(pool, xs64, beam_width=BEAM, hotwords=hot)
        dt = time.perf_counter() - t0
    return texts, frames / dt, dt


_T0 = time.perf_counter()


def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=T)
    ap.add_argument("--lm-words", type=int, default=20000)
    ap.add_argument("--lm-sentences", type=int, default=60000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="headline", choices=["headline", "config2"],
                    help="headline = BASELINE metric config; config2 = 29-char alphabet, no LM, D_flat stress (diagnostics)")
    ap.add_argument("--boost", type=float, default=6.0, help="D_words peak boost (6.0 = headline; diagnostics otherwise)")
    ap.add_argument("--phases", action="store_true", help="print the per-phase tick breakdown of utterance 0")
    ap.add_argument("--cpu-sample", type=int, default=0, help="utterances in the CPU sample (0: eight per core)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("CTC_BENCH_FORCE_DIST") == "1"  # the latter: 1-GPU rehearsal of the N>1 path
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        args.no_cpu_baseline = True  # the CPU leg runs at N=1 only (and must not fork after HIP is up)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    os.environ["CTCDEC_DEVICE"] = str(local_rank)

    from pyctcdecode_amd import build_ctcdecoder
    from pyctcdecode_amd.parallel import gather_texts

    # git-ignored but shipped to the GPU box with the snapshot (saves ~25 s of ARPA generation per run)
    cache = os.path.join(ROOT, "bench_cache") if os.access(ROOT, os.W_OK) else "/tmp/ctc_bench"
    if args.workload == "config2":  # diagnostics only: never the reported metric
        labels, hot, lm = synth.LIBRI_LABELS, None, None
        xs = [synth.d_flat(2, rank * args.batch + u, args.frames, 29) for u in range(args.batch)]
        args.no_cpu_baseline = True
    else:
        if rank == 0:
            lm, labels, hot = build_assets(cache, args.lm_words, args.lm_sentences)
        if use_dist:
            dist.barrier()
        if rank != 0:
            lm, labels, hot = build_assets(cache, args.lm_words, args.lm_sentences)
        log("assets ready")
        xs = make_batch(lm, labels, rank * args.batch, args.batch, args.frames, args.boost)
    log("batch generated")
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # timed BEFORE the HIP runtime exists in this process: forking a pool afterwards is unsafe
        cores = os.cpu_count() or 1
        n_s = min(args.cpu_sample or 4 * cores, len(xs))
        ref_texts, cpu_fps, cpu_dt = cpu_baseline(lm, labels, hot, xs[:n_s], cores)
        cpu = (cores, n_s, ref_texts, cpu_fps, cpu_dt)
        log("cpu baseline: %.0f frames/s on %d cores (%d utterances, %.1f s)" % (cpu_fps, cores, n_s, cpu_dt))
    if use_dist:
        dist.barrier()
    torch.cuda.set_device(local_rank)
    decoder = build_ctcdecoder(labels, lm.path if lm is not None else None)
    log("decoder built")
    # one padded [B, T, V] tensor, the way an acoustic model hands its logits over
    dev = torch.from_numpy(np.stack(xs)).cuda()
    log("logits on device")
    total_frames = args.batch * args.frames * world

    def step():
        texts = decoder.decode_batch(None, dev, beam_width=BEAM, hotwords=hot)
        if use_dist:
            texts = gather_texts(texts)
        return texts

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
        log("warmup step done: prune %.2f ms, beam %.2f ms, native call %.2f ms" % decoder.last_timing_ms)
    prune_ms, beam_ms, call_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        texts = step()
        prune_ms.append(decoder.last_timing_ms[0])
        beam_ms.append(decoder.last_timing_ms[1])
        call_ms.append(decoder.last_timing_ms[2])
    fence()
    dt = time.perf_counter() - t0
    log("timed steps done: %.1f ms/step (prune %.2f, beam %.2f, native call %.2f)" % (
        1000 * dt / args.steps, np.mean(prune_ms), np.mean(beam_ms), np.mean(call_ms)))
    if args.phases and rank == 0:
        import ctypes as C

        lib = decoder._lib
        lib.dll.ctcdec_profile_phases(decoder._handle, 1, None, 0)
        step()
        ticks = (C.c_uint64 * 24)()
        lib.dll.ctcdec_profile_phases(decoder._handle, 0, ticks, 24)
        names = ["load", "modes", "completions(bar)", "keys", "merge", "score(bar)", "clear", "sort.rank", "rebuild.tail",
                 "rest", "finalise", "comp.src", "comp.probe", "comp.store", "score.fold", "score.probe", "score.push",
                 "sort.zero", "sort.compact", "rebuild.hist", "rebuild.dup", "rebuild.build", "comp.syncmem", "pool.prune"]
        if getattr(decoder, "last_beam_kernel", 0) == 1:
            names = ["modes", "comp.end", "keys", "match", "fold", "score", "rank", "build.write", "finalise", "compact",
                     "push", "prefetch_tok", "build.gather", "fetch", "comp.begin"] + ["-"] * 9
        tot = float(sum(ticks)) or 1.0
        log("phase ticks (utterance 0, 100 MHz): " + ", ".join(
            "%s %.0f us (%.0f%%)" % (n, t / 100.0, 100.0 * t / tot) for n, t in zip(names, ticks) if t))
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = 1000.0 * dt / args.steps
        value = total_frames * args.steps / dt
        frames_per_launch = args.batch * args.frames
        beam_avg = float(np.mean(beam_ms)) if beam_ms else 0.0
        prune_avg = float(np.mean(prune_ms)) if prune_ms else 0.0
        algo_bytes = 4.0 * V * frames_per_launch  # 4*V bytes per logit frame
        dominant = "beam_decode" if beam_avg >= prune_avg else "frame_prune"
        dom_ms = max(beam_avg, prune_avg)
        achieved = algo_bytes / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
        traffic = None
        traffic_src = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if os.path.exists(pmc_path) and args.batch == 512 and args.frames == T:
            # PMC counters cannot be collected from inside this process; they come from the committed
            # rocprofv3 --pmc passes of this very command (profiles/r01_pmc_hbm_traffic.json)
            with open(pmc_path) as f:
                pmc = json.load(f)["kernels"]
            k = pmc["beam_decode<128,256>"] if dominant == "beam_decode" else pmc["frame_prune_f32x4<4>"]
            traffic = (k.get("fetch_bytes_corrected") or k.get("fetch_bytes_raw", 0)) + 1024.0 * k["WRITE_SIZE_KiB_raw"]
            traffic_src = "profiles/r01_pmc_hbm_traffic.json (FETCH_SIZE + WRITE_SIZE per launch)"
        out = {
            "metric": "logit-frames/sec (whole node) at beam=100, V=1024, 4-gram LM",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[3] per-GPU shard: BPE V=1024, synthetic 4-gram ARPA (%d words) + 25 hot "
                            "words, beam=100, decode_batch of %d utterances x T=%d, D_words(boost 6.0), fp32 logits "
                            "resident in HBM" % (args.lm_words, args.batch, args.frames),
                "utterances_per_gpu": args.batch,
                "frames_per_utterance": args.frames,
                "vocab": V,
                "beam_width": BEAM,
                "parallelism": "utterance shard per GPU, RCCL all_gather of texts" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms": dom_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "stages_ms": {"frame_prune": prune_avg, "beam_decode": beam_avg,
                          "frame_prune_GBps": (algo_bytes / (prune_avg * 1e-3) / 1e9) if prune_avg > 0 else None},
        }
        if cpu is not None:
            cores, n_s, ref_texts, cpu_fps, cpu_dt = cpu
            out["cpu_baseline"] = {
                "value": cpu_fps,
                "unit": "frames/s",
                "cores": cores,
                "kind": "port",
                "sample": "oracle decode_batch (pure-Python port of the reference) on a fork Pool(%d): first %d "
                          "utterances x T=%d of the same batch, %.1f s wall" % (cores, n_s, args.frames, cpu_dt),
                "texts_match_gpu": ref_texts == texts[:n_s],
            }
            out["speedup_vs_cpu_port"] = value / cpu_fps if cpu_fps > 0 else None
        if use_dist:
            # RCCL prints its NCCL_DEBUG=VERSION banner through C stdio: push it out first so that the JSON
            # line is the last thing on stdout
            import ctypes

            ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
