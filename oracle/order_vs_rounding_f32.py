"""ORACLE-SIDE EVIDENCE (test infrastructure; needs oracle/_ref, i.e. the unmodified reference staged by oracle/make_ref.py).

Question ("float32 strict-order residue"): the product returns `libri_f32_notprob_tiny` (and only it, of
1 249 beam lists) in another ORDER than the committed golden inside runs of EXACTLY equal reference scores. For float32
input the reference computes `_log_softmax` in float32 (decoder.py:180-197): numpy's float32 `exp`, a float32 pairwise
sum and numpy's float32 `log`. numpy implements float32 exp / log TWICE: a hand-written SIMD kernel (Cody-Waite reduction
+ a rational polynomial, `loops_exponent_log.dispatch.c.src`) that runs where AVX512F or AVX2+FMA3 exist -- the machine
the goldens were generated on -- and the C library's `expf` / `logf` everywhere else (other x86 builds, every ARM host,
`NPY_DISABLE_CPU_FEATURES`). The two differ in the last bit for a few per cent of arguments.

Experiment: run the UNMODIFIED reference on every float32 golden input twice -- once as the goldens were made, once in a
child process whose numpy is told to leave its AVX512F / AVX2 / FMA3 kernels alone (NPY_DISABLE_CPU_FEATURES; nothing else
changes: same numpy, same reference, same input bytes) -- and compare the returned beam ORDER.

    python oracle/order_vs_rounding_f32.py            (prints the cases whose order the reference itself does not keep)

Result on the container the goldens come from (numpy 2.2.6, AVX512F present): see the tail of this docstring's twin,
profiles/r06_order_vs_rounding_f32.txt -- `libri_f32_notprob_tiny` is among them: its order inside runs of equal scores is
a property of the CPU numpy ran on, not of the algorithm. The product's float32 route (its own float32 exponential; scores
within 2e-5 of either numpy) therefore matches the reference to the extent the reference matches itself.
"""
import json
import logging
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DISABLE = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"


def _float32_cases(full: bool):
    """(name, labels, arpa path, unigrams, build kwargs, float32 input, decode kwargs) of every float32 golden."""
    import gzip

    import numpy as np

    import synth
    from tests.golden_util import GOLD, TOY_ARPA

    out = []
    with open(os.path.join(GOLD, "cases_probs.json")) as f:
        probs = json.load(f)["cases"]
    inputs = np.load(os.path.join(GOLD, "inputs_probs.npz"))
    toy_labels = [" ", "b", "g", "n", "s", "u", "y", ""]
    for c in probs:
        if c["dtype"] != "float32":
            continue
        labels = synth.LIBRI_LABELS if c["labels"] == "libri" else toy_labels
        out.append((c["name"], list(labels), TOY_ARPA if c["lm"] else None, None, {}, inputs[c["name"]], dict(c["decode"])))
    if full:
        import bench

        cache = os.path.join(ROOT, "bench_cache") if os.access(ROOT, os.W_OK) else "/tmp/ctc_bench"
        lm, labels, hot = bench.build_assets(cache, 20000, 60000)
        for path in ("cases_full.json.gz", "cases_peaky.json.gz"):
            with gzip.open(os.path.join(GOLD, path), "rt", encoding="utf-8") as f:
                cases = json.load(f)["cases"]
            for c in cases:
                if c["dtype"] != "float32":
                    continue
                kw = dict(c["decode"])
                if c["kind"] == "config2":
                    x = synth.d_flat(2, c["utt"], c["frames"], 29)
                    out.append((c["name"], list(synth.LIBRI_LABELS), None, None, {}, x.astype(np.float32), kw))
                    continue
                if c["kind"] == "peaky":
                    x = synth.d_peaky(bench.CONFIG_ID + 1, c["utt"], bench.T, labels, True, lm.words, lm.sentences, len(labels),
                                      boost=c["boost"], unsure=c["unsure"])
                else:
                    x = synth.d_words(bench.CONFIG_ID, c["utt"], bench.T, labels, True, lm.words, lm.sentences, len(labels), boost=6.0)
                kw["hotwords"] = hot if kw.get("hotwords") == "bench" else None
                out.append((c["name"], list(labels), lm.path, None, {}, x.astype(np.float32), kw))
    return out


def _worker(full: bool) -> None:
    import numpy as np

    from oracle import make_ref

    ref = make_ref.import_reference()
    logging.disable(logging.CRITICAL)
    res = {}
    # which float32 exp is this process running? (a probe: the SIMD kernel and libm differ on this argument set)
    probe = np.exp(np.linspace(-20, 0, 4001, dtype=np.float32))
    res["__exp_probe__"] = int(probe.view(np.uint32).astype(np.uint64).sum())
    for name, labels, arpa, unigrams, build, x, kw in _float32_cases(full):
        dec = ref.build_ctcdecoder(labels, arpa, unigrams, **build)
        with np.errstate(all="ignore"):
            beams = dec.decode_beams(x, **kw)
        res[name] = [[b.text, [[w, int(s), int(e)] for w, (s, e) in b.text_frames], float(b.logit_score), float(b.lm_score)]
                     for b in beams]
        dec.cleanup()
    json.dump(res, sys.stdout)


def run(full: bool = False):
    """{case: (order differs, beams whose scores differ, largest |lm_score difference|)} between numpy's SIMD float32 exp / log and
    the C library's, on the unmodified reference; plus the two probe sums (different = the two processes really ran different exps)."""
    outs = []
    for disable in (None, DISABLE):
        env = dict(os.environ)
        env["PYTHONDONTWRITEBYTECODE"] = "1"
        env.pop("NPY_DISABLE_CPU_FEATURES", None)
        if disable:
            env["NPY_DISABLE_CPU_FEATURES"] = disable
        cmd = [sys.executable, os.path.abspath(__file__), "--worker"] + (["--full"] if full else [])
        p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            raise RuntimeError(p.stderr[-2000:])
        outs.append(json.loads(p.stdout))
    a, b = outs
    report = {}
    for name in a:
        if name.startswith("__"):
            continue
        ka = [(t, json.dumps(f)) for t, f, _, _ in a[name]]
        kb = [(t, json.dumps(f)) for t, f, _, _ in b[name]]
        same_set = sorted(ka) == sorted(kb)
        sb = {k: v for k, v in zip(kb, b[name])}
        gaps = [abs(va[3] - sb[k][3]) for k, va in zip(ka, a[name]) if k in sb]
        report[name] = {"beams": len(ka), "order_differs": ka != kb, "same_beam_set": same_set,
                        "first_difference_at_rank": next((i for i, (p, q) in enumerate(zip(ka, kb)) if p != q), None),
                        "beams_with_different_lm_score": sum(1 for g in gaps if g != 0.0), "max_lm_score_gap": max(gaps) if gaps else 0.0}
    return report, a["__exp_probe__"], b["__exp_probe__"]


if __name__ == "__main__":
    if "--worker" in sys.argv:
        _worker("--full" in sys.argv)
        sys.exit(0)
    import numpy as np

    report, pa, pb = run("--full" in sys.argv)
    print("numpy %s; float32 exp probe sums: default dispatch %d, with NPY_DISABLE_CPU_FEATURES=\"%s\" %d (%s)"
          % (np.__version__, pa, DISABLE, pb, "different kernels ran" if pa != pb else "THE SAME kernel ran: no evidence"))
    for name, r in report.items():
        print("%-28s %3d beams  order %s  (first difference at rank %s; same beam set: %s; %d beams' lm_score differ, max %.3g)"
              % (name, r["beams"], "DIFFERS" if r["order_differs"] else "equal", r["first_difference_at_rank"], r["same_beam_set"],
                 r["beams_with_different_lm_score"], r["max_lm_score_gap"]))
    unstable = [n for n, r in report.items() if r["order_differs"]]
    print("%d of %d float32 golden cases: the unmodified reference returns another beam order when numpy's float32 exp / log "
          "come from the C library instead of its SIMD kernels: %s" % (len(unstable), len(report), ", ".join(unstable) or "-"))
