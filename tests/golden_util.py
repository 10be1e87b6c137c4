"""Helpers shared by the parity tests: load tests/golden/* and rebuild the seeded LM files."""
import json
import os

import numpy as np

import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LM_DIR = os.path.join(GOLD, "_lm")
TOY_ARPA = os.path.join(GOLD, "bugs_bunny_kenlm.arpa")


def load_cases():
    with open(os.path.join(GOLD, "cases.json")) as f:
        cases = json.load(f)
    inputs = np.load(os.path.join(GOLD, "inputs.npz"))
    return cases, inputs


def load_known():
    with open(os.path.join(GOLD, "known_answers.json")) as f:
        return json.load(f)


def lm_path(spec):
    """Golden LM spec -> ARPA path (the synthetic ones are regenerated deterministically)."""
    if spec is None:
        return None
    if spec == "toy":
        return TOY_ARPA
    lm = synth.SynthLM(LM_DIR, spec["n_words"], spec["n_sent"], order=spec["order"], seed=spec["seed"],
                       upper=spec.get("upper", False))
    return lm.path


STATS = {"beam_lists": 0, "beams": 0, "max_gap": 0.0, "max_gap_what": "", "tie_runs": 0, "tie_run_beams": 0,
         # strict view of the same comparisons (round 4): how often the order really differed from the reference's
         "lists_order_differs": 0, "lists_order_differs_what": [], "exact_tie_runs": 0, "exact_tie_runs_permuted": 0,
         "near_tie_runs_permuted": 0}
STRICT = os.environ.get("CTC_STRICT_ORDER") == "1"  # any order difference fails (to list the cases; expected: the last-bit ties)


def check_beams(got, expected, tol=1e-9, what="", tie_tol=1e-9):
    """got: list of (text, frames[(word,(s,e))], logit, lm); expected: golden dicts. All bounds are ABSOLUTE.

    |logit_score - reference| and |lm_score - reference| <= tol for every beam. Order must match the reference
    exactly, EXCEPT inside runs of beams whose reference lm_scores are within ``tie_tol`` of each other: such near-ties
    are decided by the last bit of exp/log/sum rounding, which differs between libm, numpy's SIMD loops and the device
    (numpy's own AVX512 exp differs from libm's in 4.5% of inputs on this machine), so the reference itself orders them
    differently on different hosts.  Inside a run the (text, frames) multisets must still agree.
    How often the tie window was needed, and the largest score gap seen, are counted in STATS (printed at the end of a
    test session by tests/conftest.py).
    """
    assert len(got) == len(expected), "%s: %d beams, expected %d" % (what, len(got), len(expected))
    STATS["beam_lists"] += 1
    STATS["beams"] += len(got)
    _strict_view(got, expected, what, tie_tol)
    k = 0
    n = len(expected)
    while k < n:
        j = k + 1
        while j < n and abs(expected[j]["lm"] - expected[j - 1]["lm"]) <= tie_tol:
            j += 1
        gs = sorted((g[0], [[w, int(s), int(t)] for w, (s, t) in g[1]]) for g in got[k:j])
        es = sorted((e["text"], e["frames"]) for e in expected[k:j])
        assert gs == es, "%s beams %d..%d differ:\n got %r\n exp %r" % (what, k, j - 1, gs, es)
        if j - k > 1:
            STATS["tie_runs"] += 1
            STATS["tie_run_beams"] += j - k
        bound = max(tol, tie_tol) if j - k > 1 else tol
        for g, e in zip(got[k:j], expected[k:j]):
            if j - k == 1:  # (inside a tie run the logit scores are compared as a multiset below: beams that tie on
                # lm_score may split it differently between acoustics and LM / hot-word bonus)
                gap = abs(g[2] - e["logit"])
                assert gap <= bound or g[2] == e["logit"], (what, k, g[2], e["logit"])
                _note_gap(gap, what)
            gap = abs(g[3] - e["lm"])
            assert gap <= bound or g[3] == e["lm"], (what, k, g[3], e["lm"])  # (== : equal infinities)
            _note_gap(gap, what)
        if j - k > 1:  # logit scores inside a tie run: compare as multisets
            gl = sorted(g[2] for g in got[k:j])
            el = sorted(e["logit"] for e in expected[k:j])
            for a, b in zip(gl, el):
                assert abs(a - b) <= bound or a == b, (what, k, a, b)
        k = j


def _ident(text, frames):
    return (text, [[w, int(a), int(b)] for w, a, b in frames])


def _strict_view(got, expected, what, tie_tol):
    """Bookkeeping only (unless CTC_STRICT_ORDER=1): is the returned order EXACTLY the reference's, and where it is not, was
    it inside a run of exactly equal reference scores (the reference's own order there is its arrival order: a tie-break
    rule of ours would be at fault) or of scores a last-bit rounding apart (libm vs numpy vs device)?"""
    g_ids = [_ident(g[0], [[w, s, t] for w, (s, t) in g[1]]) for g in got]
    e_ids = [_ident(e["text"], e["frames"]) for e in expected]
    if g_ids != e_ids:
        STATS["lists_order_differs"] += 1
        if len(STATS["lists_order_differs_what"]) < 12:
            STATS["lists_order_differs_what"].append(what)
    n = len(expected)
    k = 0
    while k < n:
        j = k + 1
        while j < n and expected[j]["lm"] == expected[k]["lm"]:
            j += 1
        if j - k > 1:
            STATS["exact_tie_runs"] += 1
            if g_ids[k:j] != e_ids[k:j]:
                STATS["exact_tie_runs_permuted"] += 1
        k = j
    k = 0
    while k < n:
        j = k + 1
        while j < n and abs(expected[j]["lm"] - expected[j - 1]["lm"]) <= tie_tol:
            j += 1
        if j - k > 1 and g_ids[k:j] != e_ids[k:j] and any(expected[i]["lm"] != expected[k]["lm"] for i in range(k, j)):
            STATS["near_tie_runs_permuted"] += 1
        k = j
    if STRICT:
        assert g_ids == e_ids, "%s: beam order differs from the reference (CTC_STRICT_ORDER=1)" % what


def _note_gap(gap, what):
    if gap == gap and gap > STATS["max_gap"]:
        STATS["max_gap"] = gap
        STATS["max_gap_what"] = what
