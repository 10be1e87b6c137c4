"""ORACLE / CPU BASELINE (test + bench infrastructure): make the UNMODIFIED reference importable on the GPU box.

    python oracle/make_ref.py

Copies the reference package's own sources (/root/reference/pyctcdecode/*.py, unmodified, no tests or data)
into the git-ignored directory oracle/_ref/pyctcdecode/.  Nothing under oracle/_ref/ is ever committed (see
.gitignore): like the built .so files it only travels with the gpurun snapshot, so that `bench.py`'s
cpu_baseline leg can time the reference's decode_batch itself on the GPU box's host cores (kind: "reference")
instead of the oracle's restatement (kind: "port").  The reference's two missing dependencies come from
oracle/refshim/ (kenlm -> oracle/arpa_lm.py, pygtrie -> a 20-line trie).

Only bench.py's cpu_baseline leg and tests import oracle/_ref; the product never does.
"""
import filecmp
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/pyctcdecode"
DST = os.path.join(HERE, "_ref", "pyctcdecode")
FILES = ["__init__.py", "alphabet.py", "constants.py", "decoder.py", "language_model.py"]


def available() -> bool:
    return all(os.path.exists(os.path.join(DST, f)) for f in FILES)


def build(verbose: bool = True) -> bool:
    """Returns True when oracle/_ref holds the reference package afterwards."""
    if not os.path.isdir(SRC):
        return available()
    os.makedirs(DST, exist_ok=True)
    for f in FILES:
        a, b = os.path.join(SRC, f), os.path.join(DST, f)
        if not os.path.exists(b) or not filecmp.cmp(a, b, shallow=False):
            shutil.copyfile(a, b)
            if verbose:
                print("oracle/_ref/pyctcdecode/" + f, file=sys.stderr)
    return available()


def import_reference():
    """The unmodified reference package (module object), with the kenlm / pygtrie stand-ins on the path."""
    if not available():
        raise ImportError("oracle/_ref is empty (run python oracle/make_ref.py where /root/reference exists)")
    for p in (os.path.join(HERE, "refshim"), os.path.join(HERE, "_ref")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pyctcdecode  # noqa: E402  (the reference, from oracle/_ref)

    assert os.path.dirname(os.path.abspath(pyctcdecode.__file__)) == DST, pyctcdecode.__file__
    return pyctcdecode


if __name__ == "__main__":
    print("reference available:", build())
