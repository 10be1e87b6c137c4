"""Build the current tree (optionally with extra hipcc flags, e.g. -DCTC_SOMETHING) into pyctcdecode_amd/variants/libctcdec_<name>.so,
next to -- not over -- the product library, for A/B runs on one GPU box inside one process (tools/ab_bench.py lib=<path>).
  python tools/build_variant.py NAME [hipcc flags ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyctcdecode_amd import build  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    vdir = os.path.join(build.HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    build.OUT = os.path.join(vdir, "libctcdec_%s.so" % name)
    obj = os.path.join(vdir, "_obj_" + name)
    build._obj_dir = lambda: obj
    build.PYTEXTS = os.path.join(vdir, "_unused_pytexts.so")
    if flags:
        os.environ["CTCDEC_HIPCC_EXTRA"] = " ".join(flags)
    print(build.build(force=True, verbose=False))


if __name__ == "__main__":
    main()
