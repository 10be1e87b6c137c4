"""Summarise a CTCDEC_WAVE_TIMES dump of the wave kernel (n_utts x 4 uint64: start, end in 100 MHz ticks, HW_ID | XCC_ID << 32,
frames | utterance << 32): how long each wave ran, how far apart the waves of a launch finish, and whether a wave's pace
depends on where it sits (its slot on the SIMD = its age rank, SIMD, CU, XCD).   python tools/wave_times.py <file> [...]"""
import sys

import numpy as np


def summarise(path):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    hw = (a[:, 2] & 0xFFFFFFFF).astype(np.int64)
    xcc = (a[:, 2] >> 32).astype(np.int64) & 0xF
    frames = (a[:, 3] & 0xFFFFFFFF).astype(np.int64)
    wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    base = t0.min()
    dur = (t1 - t0) / 100.0  # us
    end = (t1 - base) / 100.0
    start = (t0 - base) / 100.0
    n = len(a)
    print("%s: %d waves, launch span %.2f ms; starts within %.1f us" % (path, n, end.max() / 1e3, start.max()))
    q = np.percentile(end, [0, 5, 25, 50, 75, 95, 100]) / 1e3
    print("  wave END time (ms after the first start): min %.2f  p5 %.2f  p25 %.2f  median %.2f  p75 %.2f  p95 %.2f  max %.2f" % tuple(q))
    print("  mean wave lifetime %.2f ms = %.0f %% of the span; us per frame: mean %.2f, fastest wave %.2f, slowest %.2f"
          % (dur.mean() / 1e3, 100 * dur.mean() / end.max(), (dur / frames).mean(), (dur / frames).min(), (dur / frames).max()))
    for name, key in (("slot on its SIMD (HW_ID.wave_id)", wave_id), ("SIMD", simd), ("XCD", xcc)):
        vals = sorted(set(key.tolist()))
        print("  by %s: " % name + "  ".join("%d: %.2f ms (n=%d)" % (v, end[key == v].mean() / 1e3, (key == v).sum()) for v in vals))
    # waves sharing one SIMD: rank by end time inside each (xcc, se, sh, cu, simd) group
    grp = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    grp = grp * 4 + simd
    order = np.lexsort((end, grp))
    g_sorted = grp[order]
    first = np.r_[True, g_sorted[1:] != g_sorted[:-1]]
    idx_in = np.arange(n) - np.maximum.accumulate(np.where(first, np.arange(n), 0))
    sizes = np.bincount(np.unique(g_sorted, return_inverse=True)[1])
    print("  SIMDs used: %d; waves per SIMD: min %d max %d" % (len(sizes), sizes.min(), sizes.max()))
    for r in range(int(idx_in.max()) + 1):
        sel = order[idx_in == r]
        print("    %d. wave of its SIMD to finish: mean end %.2f ms, mean slot %.2f" % (r + 1, end[sel].mean() / 1e3, wave_id[sel].mean()))
    blk = np.arange(n)
    import os
    if os.path.exists(path + ".w"):  # the weights utt_weigh gave the workgroups' utterances (relative cost of a frame)
        w = np.fromfile(path + ".w", dtype=np.float32)[:n].astype(np.float64)
        print("  weights: min %.3f  p5 %.3f  median %.3f  p95 %.3f  max %.3f" % tuple(np.percentile(w, [0, 5, 50, 95, 100])))
        for s_ in sorted(set(wave_id.tolist())):
            m = wave_id == s_
            if m.sum() > 8:
                c = np.corrcoef(w[m], dur[m])[0, 1]
                k, b0 = np.polyfit(w[m], dur[m] / 1e3, 1)
                print("    slot %d: corr(weight, lifetime) %.3f; lifetime = %.2f ms + %.2f ms x weight  => cost at weight 0 / cost at the mean weight = %.2f" % (s_, c, b0, k, b0 / (b0 + k)))
        top = np.argsort(-end)[:12]
        print("  the last 12 waves to end: end ms / weight / slot / block: " + "  ".join("%.2f/%.3f/%d/%d" % (end[i] / 1e3, w[i], wave_id[i], i) for i in top))
        # do block b and block b + 1024 share a SIMD?
        S = 1024
        if n >= 2 * S:
            same = (grp[:S] == grp[S:2 * S]).mean()
            print("  blocks b and b + %d on the same SIMD: %.1f %%" % (S, 100 * same))
    print("  by dispatch order (block index / 1024): " + "  ".join("%d: end %.2f ms, slot %.2f" % (k, end[(blk >> 10) == k].mean() / 1e3, wave_id[(blk >> 10) == k].mean()) for k in range((n + 1023) >> 10)))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarise(p)
