"""What predicts how long a wave of the beam_wave launch lives? (round 6; this container + one wave-times dump from the GPU box)
  1. 64 bench utterances through the statistics build of the simulator (tests/sim, -DCTC_STATS: live beams N, survivors ns, pool
     entries per frame) against their lifetimes as the OLDEST waves of their SIMDs in a CTCDEC_WAVE_PRIO=none launch
     (gpurun_out/.../wt_none.bin: block b decodes utterance b there, blocks 0..1023 are slot 0);
  2. what the logits alone can say: features of the prune stage's output for 256 utterances against the same lifetimes;
  3. whether an utterance's rate persists (first k frames against the rest).
  python tools/weigh_predictors.py <wt_none.bin> [n_sim_utts=64]      (the simulator part takes ~1 min per 8 utterances and core)"""
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sim_stats(first, n):
    """[(N, ns, pool, kept)] per full frame of utterances first .. first+n-1, from the -DCTC_STATS simulator's stderr"""
    code = r'''
import os, sys
sys.path.insert(0, %r)
from tests.sim import build_sim
OUT = os.path.join(build_sim.OUT_DIR, "libctcdec_sim_stats.so")
from pyctcdecode_amd import _binding as B
B._LIB = B.Library(OUT)
import bench
from pyctcdecode_amd import build_ctcdecoder
lm, labels, hot = bench.build_assets(os.path.join(%r, "bench_cache"), 20000, 60000)
dec = build_ctcdecoder(labels, lm.path)
xs = bench.make_batch(lm, labels, %d, %d, 1000, 6.0, 1)
for u in range(%d):
    sys.stderr.write("UTT %%d\n" %% (%d + u)); sys.stderr.flush()
    dec.decode(xs[u], beam_width=100, hotwords=hot)
''' % (ROOT, ROOT, first, n, n, first)
    return subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)


def build_stats_lib():
    from tests.sim import build_sim

    out = os.path.join(build_sim.OUT_DIR, "libctcdec_sim_stats.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in build_sim.DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-DCTC_SIM", "-DCTC_STATS", "-DCTC_TEXT_WIN=48",
                               "-DCTC_TEXT_LIST=16", "-Wno-unused-function", "-o", out] + build_sim.SOURCES)


def main():
    wt = sys.argv[1]
    n_sim = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    a = np.fromfile(wt, dtype=np.uint64).reshape(-1, 4)
    dur = (a[:, 1].astype(np.int64) - a[:, 0].astype(np.int64)) / 1e5  # ms
    uid = (a[:, 3] >> np.uint64(32)).astype(int)
    life = {int(u): d for u, d in zip(uid, dur)}
    build_stats_lib()
    procs = [sim_stats(k, min(8, n_sim - k)) for k in range(0, n_sim, 8)]
    utts, cur = {}, None
    for p in procs:
        for line in p.stderr:
            if line.startswith("UTT"):
                cur = int(line.split()[1])
                utts[cur] = []
            elif line.startswith("ST"):
                utts[cur].append(tuple(map(int, line.split()[1:])))
        p.wait()
    F, y = [], []
    for u in sorted(utts):
        r = np.array(utts[u])
        N, ns = r[:, 0], r[:, 1]
        passes = sum(math.ceil(s / max(1, 64 // max(n, 1))) if n <= 64 else s for n, s in zip(N, ns))
        F.append([ns.sum(), (ns ** 2).sum(), (ns[1:] * ns[:-1]).sum(), passes, (N * ns).sum(), N.sum()])
        y.append(life[u])
    F, y = np.array(F, float), np.array(y)
    names = ["sum of survivors", "sum of survivors^2", "sum of ns(t) ns(t-1)", "candidate passes", "candidates (N x ns)", "sum of live beams"]
    print("1. %d utterances, simulator statistics against their lifetimes as slot-0 waves (mean %.2f ms, std %.2f):" % (len(y), y.mean(), y.std()))
    for j, nm in enumerate(names):
        print("   %-22s r = %.3f" % (nm, np.corrcoef(F[:, j], y)[0, 1]))
    c = np.corrcoef(F[:, 4], y)[0, 1]
    b = c * y.std() / F[:, 4].std()
    print("   lifetime = %.2f ms + %.3f us per candidate (the average frame: %.1f candidates)" % (y.mean() - b * F[:, 4].mean(), 1e3 * b, F[:, 4].mean() / 1000))
    allr = np.array([t for u in utts for t in utts[u]])
    N, ns = np.maximum(allr[:, 0], 1), allr[:, 1]
    small = N <= 64
    cur = np.where(small, np.ceil(ns / np.maximum(1, 64 // N)), ns).sum()
    dense = np.where(small, np.ceil(ns * N / 64), ns).sum()
    two = np.where(small, np.ceil(ns / np.maximum(1, 128 // N)), ns).sum()
    f = float(len(allr))
    print("   passes per frame: whole labels per 64-lane pass (the kernel) %.2f; candidates packed densely %.2f; 128 candidates per pass "
          "%.2f; live beams mean %.1f (median %d), survivors %.2f, lanes in use %.0f %%" % (cur / f, dense / f, two / f, N.mean(), int(np.median(N)),
                                                                                             ns.mean(), 100 * (N * ns).sum() / (cur * 64)))
    print("3. does an utterance's rate persist? mean cost per frame (85 + candidates) of the first k frames against the rest:")
    for k in (32, 64, 128, 256, 500):
        A, B_ = [], []
        for u in sorted(utts):
            r = np.array(utts[u])
            cst = 85 + r[:, 0] * r[:, 1]
            A.append(cst[:k].mean())
            B_.append(cst[k:].mean())
        print("   k = %3d: r = %.3f" % (k, np.corrcoef(A, B_)[0, 1]))
    # 2. the logits alone
    import bench

    lm, labels, hot = bench.build_assets(os.path.join(ROOT, "bench_cache"), 20000, 60000)
    n = 256
    xs = bench.make_batch(lm, labels, 0, n, 1000, 6.0, 8).astype(np.float64)
    yy = dur[:n]
    m = xs.max(-1, keepdims=True)
    lp = xs - m - np.log(np.exp(xs - m).sum(-1, keepdims=True))
    srt = -np.sort(-lp, axis=-1)[:, :, :16]
    nsv = (lp >= -5).sum(-1)
    feats = {"sum of survivors": nsv.sum(1), "sum of survivors^2": (nsv ** 2).sum(1), "labels within 1 of the best": (srt >= srt[:, :, :1] - 1).sum((1, 2)),
             "labels within 3 of the best": (srt >= srt[:, :, :1] - 3).sum((1, 2)), "sum of the best log-prob": srt[:, :, 0].sum(1),
             "sum of the gap best - second": (srt[:, :, 0] - srt[:, :, 1]).sum(1), "sum of the second's probability": np.exp(srt[:, :, 1]).sum(1),
             "sum of the frame entropies": -(np.exp(srt) * srt).sum((1, 2))}
    print("2. %d utterances, what the logits alone say, against the same lifetimes:" % n)
    for k, v in feats.items():
        print("   %-32s r = %+.3f" % (k, np.corrcoef(v.astype(float), yy)[0, 1]))


if __name__ == "__main__":
    main()
