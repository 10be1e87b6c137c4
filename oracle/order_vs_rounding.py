"""ORACLE-SIDE EVIDENCE (test infrastructure; needs oracle/_ref, i.e. the unmodified reference staged by oracle/make_ref.py).

Question ("beam order is not bit-exact"): when this package returns the reference's beams in another
ORDER inside a run of equal or nearly equal lm_scores, is a tie-break rule wrong -- or is that order not a property of the
algorithm at all?

Experiment: run the UNMODIFIED reference twice on the same float64 input, the second time with ONE thing changed -- the
normaliser of its log-softmax (decoder.py:180-197) summed exactly (math.fsum) instead of pairwise (np.sum): a change of
at most the last bit of a frame's log-probabilities, and the mathematically better value. Every golden case whose beam
order the reference itself does not keep under that change is a case whose order hinges on last-bit rounding.

    python oracle/order_vs_rounding.py            (prints the unstable cases; round 4: exactly toy_nolm_16beams,
                                                   toy_history_prune, toy_lm_unk0_prune20, toy_lm_autounigrams_prune60 --
                                                   the four cases in which the round-3 kernels' order differed)

The other half of the answer is constructive: with the normaliser summed in numpy's own order (csrc/np_sum.h) the
simulator build returns all 58 golden cases bit for bit -- scores and order (tests/test_order_stability.py)."""
import logging
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _exact_normaliser(orig):
    def log_softmax(x, axis):
        x = np.asarray(x)
        if x.size == 0 or x.dtype != np.float64:
            return orig(x, axis)
        x_max = np.amax(x, axis=axis, keepdims=True)
        if x_max.ndim > 0:
            x_max[~np.isfinite(x_max)] = 0
        tmp = x - x_max
        e = np.exp(tmp)
        s = np.array([[math.fsum(row)] for row in e])
        with np.errstate(divide="ignore"):
            return tmp - np.log(s)

    return log_softmax


def unstable_cases(names=None):
    """Names of the float64 golden cases (tests/golden/cases.json) whose beam order the reference does not keep."""
    from oracle import make_ref
    from tests.golden_util import lm_path, load_cases

    ref = make_ref.import_reference()
    import pyctcdecode.decoder as D  # the reference (oracle/_ref)

    cases, inputs = load_cases()
    orig = D._log_softmax
    out = []
    logging.disable(logging.CRITICAL)
    try:
        for case in cases:
            x = inputs[case["input"]]
            if x.dtype != np.float64 or x.shape[0] == 0 or (names is not None and case["name"] not in names):
                continue
            res = []
            for variant in (orig, _exact_normaliser(orig)):
                D._log_softmax = variant
                dec = ref.build_ctcdecoder(list(case["labels"]), lm_path(case["lm"]), case["unigrams"], **case["build"])
                with np.errstate(all="ignore"):
                    beams = dec.decode_beams(x, **case["decode"])
                res.append([(b.text, tuple(b.text_frames)) for b in beams])
                dec.cleanup()
            if res[0] != res[1]:
                assert sorted(res[0]) == sorted(res[1]), case["name"]  # the same beams, another order
                out.append(case["name"])
    finally:
        D._log_softmax = orig
        logging.disable(logging.NOTSET)
    return out


if __name__ == "__main__":
    names = unstable_cases()
    print("%d golden cases in which the reference itself returns another beam order when only the rounding of its "
          "log-softmax normaliser changes: %s" % (len(names), ", ".join(names)))
