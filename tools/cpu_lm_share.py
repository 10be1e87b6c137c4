"""How much of the CPU baseline is the Python stand-in for kenlm? bench.py times the UNMODIFIED reference with
oracle/arpa_lm.py in place of the C++ library (absent here). This script profiles the reference's decode() of bench
utterances (cProfile) and reports the share of the time spent inside the stand-in's entry points (BaseScore,
BeginSentenceWrite, NullContextWrite, __contains__, State()): with the real kenlm that share shrinks towards zero, so
    reference-with-real-kenlm  <=  measured rate / (1 - share)
is the bound the bench line quotes next to its CPU figures.   python tools/cpu_lm_share.py [n_utts]   (CPU only)"""
import cProfile
import logging
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import synth  # noqa: E402


def lm_share(n_utts=2, frames=None, real_posterior=False):
    """(share of decode time inside the kenlm stand-in, profiled frames/s, un-profiled frames/s) on bench utterances."""
    from oracle import make_ref

    ref = make_ref.import_reference()
    cache = os.path.join(ROOT, "bench_cache") if os.access(ROOT, os.W_OK) else "/tmp/ctc_bench"
    lm, labels, hot = bench.build_assets(cache, 20000, 60000)
    logging.disable(logging.WARNING)
    dec = ref.build_ctcdecoder(list(labels), lm.path)
    T = frames or bench.T
    if real_posterior:
        xs = [synth.d_peaky(bench.CONFIG_ID + 1, u, T, labels, True, lm.words, lm.sentences, len(labels)).astype(np.float64) for u in range(n_utts)]
    else:
        xs = [synth.d_words(bench.CONFIG_ID, u, T, labels, True, lm.words, lm.sentences, len(labels), boost=6.0).astype(np.float64)
              for u in range(n_utts)]
    t0 = time.perf_counter()
    for x in xs:
        dec.decode(x, beam_width=bench.BEAM, hotwords=hot)
    plain = sum(x.shape[0] for x in xs) / (time.perf_counter() - t0)
    pr = cProfile.Profile()
    pr.enable()
    for x in xs:
        dec.decode(x, beam_width=bench.BEAM, hotwords=hot)
    pr.disable()
    st = pstats.Stats(pr)
    total = st.total_tt
    inside = 0.0
    entry = ("BaseScore", "BeginSentenceWrite", "NullContextWrite", "__contains__", "__init__")
    for (fn, _line, name), (_cc, _nc, _tt, ct, callers) in st.stats.items():
        if fn.endswith("arpa_lm.py") and name in entry:
            # cumulative time of an entry point, counted only for calls that come from OUTSIDE the stand-in
            for (cfn, _cl, _cn), (_c1, _c2, _ctt, cct) in callers.items():
                if not cfn.endswith("arpa_lm.py"):
                    inside += cct
    logging.disable(logging.NOTSET)
    return inside / total, sum(x.shape[0] for x in xs) / total, plain


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for real in (False, True):
        share, prof_fps, fps = lm_share(n, real_posterior=real)
        print("%s input, %d utterances x %d frames: %.1f %% of the reference's decode() is spent inside the kenlm stand-in "
              "(oracle/arpa_lm.py); one process %.0f frames/s (%.0f under the profiler); with a zero-cost kenlm at most %.0f frames/s (x %.3f)"
              % ("real-posterior-like" if real else "bench (D_words boost 6)", n, bench.T, 100 * share, fps, prof_fps, fps / (1 - share), 1 / (1 - share)))
