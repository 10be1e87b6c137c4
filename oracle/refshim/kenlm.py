"""ORACLE SHIM (test infrastructure, this container only): stand-in for the `kenlm` module.

The reference imports `kenlm` (/root/reference/pyctcdecode/decoder.py:55-61,
language_model.py:28-34); it is not installed here and there is no network.  This shim exposes
the five calls the reference makes, backed by oracle/arpa_lm.py.  It exists only so the
UNMODIFIED reference can be imported to generate golden vectors (oracle/make_golden.py).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from arpa_lm import ArpaModel, ArpaState  # noqa: E402


class State(ArpaState):
    pass


class Model(ArpaModel):
    pass
