"""Shared by the CPU (sim) and GPU tests: the frame-prune stage's survivor lists against CPython's own set."""
import ctypes as C

import numpy as np

from oracle.ctc_oracle import normalise_logits


def survivors(dec, x, token_min_logp):
    """-> list of (ids, logps) per frame, straight from ctcdec_frame_survivors."""
    x = np.ascontiguousarray(x)
    T, V = x.shape
    dtype = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.float16): 2}[x.dtype]
    counts = (C.c_int32 * max(T, 1))()
    ids = (C.c_int32 * max(T * V, 1))()
    lps = (C.c_double * max(T * V, 1))()
    dec._lib.check(dec._lib.dll.ctcdec_frame_survivors(dec._handle, x.ctypes.data_as(C.c_void_p), T, dtype, 0,
                                                        float(token_min_logp), V, counts, ids, lps))
    return [([ids[t * V + k] for k in range(counts[t])], [lps[t * V + k] for k in range(counts[t])]) for t in range(T)]


def check_against_cpython(dec, x, token_min_logp, tol):
    got = survivors(dec, x, token_min_logp)
    import os

    # float32 rows: the product restates the reference's float32 log-softmax (the default, csrc/np_f32.h) -- the oracle's
    # normalisation on the float32 matrix itself is then the exact expectation; otherwise the exact float64 upcast
    exact32 = x.dtype == np.float32 and os.environ.get("CTCDEC_PRUNE_EXP", "np")[0] == "n"
    with np.errstate(all="ignore"):
        lp = normalise_logits(x if exact32 else x.astype(np.float64)).astype(np.float64)
    n_border = 0
    for t, (ids, lps) in enumerate(got):
        row = lp[t]
        # labels whose log-prob sits within tol of the threshold may fall on either side (last-bit exp/log)
        sure = set(int(i) for i in np.where(row >= token_min_logp + tol)[0])
        maybe = set(int(i) for i in np.where(row >= token_min_logp - tol)[0])
        amax = int(np.argmax(row))
        assert sure <= set(ids) <= (maybe | {amax}), (t, ids)
        if abs(row[amax] - token_min_logp) <= tol:  # is argmax in the where-list? undecidable at this tolerance
            n_border += 1
            continue
        # the where-list as the device saw it (borderline labels taken as they fell), argmax only if it passes
        members = set(ids) if row[amax] >= token_min_logp else set(ids) - {amax}
        if members != set(int(i) for i in np.where(row >= token_min_logp)[0]):
            n_border += 1
        # decoder.py:444-445: the set is built from the ascending index array, then `| {argmax}`
        expect = list(set(np.array(sorted(members), dtype=np.int64)) | {np.int64(amax)})
        assert ids == [int(i) for i in expect], (t, ids, expect)
        for i, v in zip(ids, lps):
            assert abs(v - row[i]) <= tol * max(1.0, abs(row[i])), (t, i, v, row[i])
    return n_border
