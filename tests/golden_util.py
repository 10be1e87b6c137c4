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


def check_beams(got, expected, tol=1e-9, what=""):
    """got: list of (text, frames[(word,(s,e))], logit, lm); expected: golden dicts."""
    assert len(got) == len(expected), "%s: %d beams, expected %d" % (what, len(got), len(expected))
    for k, (g, e) in enumerate(zip(got, expected)):
        assert g[0] == e["text"], "%s beam %d text %r != %r" % (what, k, g[0], e["text"])
        gf = [[w, int(s), int(t)] for w, (s, t) in g[1]]
        assert gf == e["frames"], "%s beam %d frames %r != %r" % (what, k, gf, e["frames"])
        assert abs(g[2] - e["logit"]) <= tol * max(1.0, abs(e["logit"])), (what, k, g[2], e["logit"])
        assert abs(g[3] - e["lm"]) <= tol * max(1.0, abs(e["lm"])), (what, k, g[3], e["lm"])
