"""Random differential hunt: the sequential-sim build of the device code (or, with FUZZ_BACKEND=hip on a GPU box,
the HIP build) through the full Python shell and C ABI against the oracle, over random vocabularies / language models (single and multi) / hot words / decode
arguments / input styles / chunkings.  TEST INFRASTRUCTURE (CPU only):  python tools/fuzz_sim_vs_oracle.py [n] [seed]
"""
import math
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logging  # noqa: E402

import synth  # noqa: E402
from oracle.arpa_lm import ArpaModel  # noqa: E402
from oracle.ctc_oracle import LMOracle, MultiLMOracle, OracleDecoder, load_unigrams_from_arpa  # noqa: E402
from pyctcdecode_amd import _binding as B  # noqa: E402
from tests.golden_util import LM_DIR, TOY_ARPA, check_beams  # noqa: E402
from tests.sim.build_sim import build  # noqa: E402

HIP = os.environ.get("FUZZ_BACKEND") == "hip"  # the product library on a GPU box instead of the sequential sim
TOL = 1e-9  # absolute (tests/golden_util.check_beams), sim and HIP alike
TOL_F32 = 1e-4 if (HIP and os.environ.get("CTCDEC_PRUNE_EXP") != "f64") else None  # float32 rows, HIP build: the packed
# float32 exponential of frame_prune (the reference's own precision for float32 input); None: fp64 like everything else


def use_backend():
    """Command-line use: pick the library (tests choose it themselves before calling run_many)."""
    if not HIP:
        B._LIB = B.Library(os.environ.get("FUZZ_LIB") or build())  # FUZZ_LIB: e.g. an AddressSanitizer build of the sim
from pyctcdecode_amd.alphabet import Alphabet  # noqa: E402
from pyctcdecode_amd.decoder import BeamSearchDecoderCTC  # noqa: E402
from pyctcdecode_amd.language_model import HotwordScorer, LanguageModel, MultiLanguageModel, NgramModel  # noqa: E402

WORDS = synth.make_words(300, seed=2)
LM_SPECS = [(300, 400, 4, 2), (200, 300, 3, 3), (200, 300, 2, 1)]


def lm_file(rng):
    if rng.random() < 0.25:
        return TOY_ARPA
    nw, ns, order, seed = LM_SPECS[int(rng.integers(0, len(LM_SPECS)))]
    return synth.SynthLM(LM_DIR, nw, ns, order=order, seed=seed).path


def one_case(rng):
    kind = rng.choice(["char", "toy", "bpe_small", "bpe255", "bpe1023"])
    if kind == "char":
        labels = list(synth.LIBRI_LABELS)
    elif kind == "toy":
        labels = [" ", "b", "g", "n", "s", "u", "y", ""]
    elif kind == "bpe_small":
        labels = ["<unk>", "▁", "a", "b", "▁a", "▁b", "▁bu", "gs", "nny", "▁bugs", "n", "y", "s", "g", "u"]
    else:
        labels = synth.make_bpe_vocab(WORDS, size=255 if kind == "bpe255" else 1023)
    alpha = Alphabet.build_alphabet(labels)
    n_lm = int(rng.choice([0, 1, 1, 1, 2, 3]))
    members, olms = [], []
    for _ in range(n_lm):
        path = lm_file(rng)
        r = rng.random()
        uni = None if r < 0.15 else (sorted(load_unigrams_from_arpa(path))[: int(rng.integers(1, 80))] if r < 0.4
                                     else sorted(load_unigrams_from_arpa(path)))
        kw = dict(alpha=float(rng.choice([0.5, 0.0, 1.0, 0.7])), beta=float(rng.choice([1.5, 0.0, 3.0])),
                  unk_score_offset=float(rng.choice([-10.0, 0.0, -4.0])), score_boundary=bool(rng.random() < 0.7))
        members.append(LanguageModel(NgramModel(path), uni, **kw))
        olms.append(LMOracle(ArpaModel(path), uni, kw["alpha"], kw["beta"], kw["unk_score_offset"], kw["score_boundary"]))
    lm = None if n_lm == 0 else (members[0] if n_lm == 1 else MultiLanguageModel(members))
    olm = None if n_lm == 0 else (olms[0] if n_lm == 1 else MultiLMOracle(olms))
    dec = BeamSearchDecoderCTC(alpha, lm)
    orc = OracleDecoder(alpha.labels, alpha.is_bpe, olm)
    V = len(alpha.labels)
    heavy = os.environ.get("FUZZ_HEAVY") == "1"  # long utterances, flat rows, wide beams: many candidates per frame
    T = int(rng.integers(0, 45)) if not heavy else int(rng.integers(40, 160))
    style = rng.choice(["normal", "peaky", "int", "prob", "masked", "words", "posterior", "posterior"]) if not heavy else rng.choice(["normal", "words", "posterior"])
    if style == "normal":
        x = rng.standard_normal((T, V)) * rng.choice([1.0, 1.5, 3.0])
    elif style == "peaky":
        x = rng.standard_normal((T, V))
        if T:
            x[np.arange(T), rng.integers(0, V, size=T)] += 6.0
    elif style == "posterior":
        # trained-CTC-like: long runs of frames with ONE survivor (the blank, or a held label) -- the single-label runs
        # of the wave kernel -- broken by frames with a competitor; sometimes quantised so that rounding matters
        x = rng.standard_normal((T, V))
        if T:
            ali = np.empty(T, dtype=np.int64)
            cur = V - 1
            for t in range(T):
                if rng.random() < 0.4:
                    cur = int(rng.integers(0, V)) if rng.random() < 0.6 else V - 1
                ali[t] = cur
            x[np.arange(T), ali] += float(rng.choice([10.0, 14.0, 20.0]))
            for t in np.nonzero(rng.random(T) < 0.15)[0]:
                x[t, int(rng.integers(0, V))] = x[t, ali[t]] - rng.uniform(0.0, 4.0)
            if rng.random() < 0.3:
                x = np.round(x * 4.0) / 4.0 + 1e-3 * rng.standard_normal((T, V))
    elif style == "int":
        # integer logits make every score difference an exact integer: massive ties. Exact ties are covered by the
        # golden vectors; here a near-tie (1 ulp) straddling the beam-width cut would be decided by the last bit
        # of exp/log, which legitimately differs between libm and numpy -- so the ties are broken by small noise
        x = rng.integers(-8, 1, size=(T, V)).astype(np.float64) + 1e-3 * rng.standard_normal((T, V))
    elif style == "prob":
        e = np.exp(rng.standard_normal((T, V)) * 2)
        x = e / e.sum(axis=1, keepdims=True) if T else e
    elif style == "masked":
        x = rng.standard_normal((T, V)) * 2
        x[:, rng.random(V) < 0.3] = -np.inf
        if T:
            x[np.arange(T), rng.integers(0, V, size=T)] = 3.0  # at least one finite entry per row
    else:
        lm_a = synth.SynthLM(LM_DIR, 300, 400, order=4, seed=2)
        space = " " if " " in alpha.labels else "|"
        try:
            x = synth.d_words(2, int(rng.integers(0, 1000)), max(T, 1), labels, alpha.is_bpe, lm_a.words, lm_a.sentences,
                              V - 1, boost=float(rng.choice([4.0, 6.0])), space_label=space).astype(np.float64)[:T]
        except Exception:
            x = rng.standard_normal((T, V))
    dt = rng.choice(["f64", "f32", "f16"])
    if dt == "f32":
        x = x.astype(np.float32)
    elif dt == "f16" and style not in ("prob",):
        x = np.clip(x, -60000, 60000).astype(np.float16)
    hot = None
    if rng.random() < 0.4:
        hot = [str(s) for s in rng.choice(["bugs", "bunny", "bun", "ab", "bugs bunny", "a", "zq", WORDS[3], WORDS[7]], size=3)]
    dkw = dict(beam_width=int(rng.choice([1, 3, 10, 25, 100, 200] if not heavy else [64, 100, 128, 200, 256])),
               beam_prune_logp=float(rng.choice([-3.0, -10.0, -30.0] if not heavy else [-10.0, -30.0])),
               token_min_logp=float(rng.choice([-5.0, -3.0, -8.0, 0.0] if not heavy else [-5.0, -8.0])),
               prune_history=bool(rng.random() < 0.5),
               hotwords=hot, hotword_weight=float(rng.choice([10.0, 3.0])))
    return dec, orc, x, dkw


def _prune_mode():
    return os.environ.get("CTCDEC_PRUNE_EXP", "np")[0]


def _oracle_input(x):
    """float32 logits go to the oracle in their own dtype when the product computes them in the reference's float32 arithmetic
    (the default); everything else as the exact float64 upcast."""
    if x.dtype == np.float32 and _prune_mode() == "n":
        return x
    return x.astype(np.float64)


def _parted_at_a_tie(dec, orc, x, dkw, win):
    """Frame by frame through partial_decode_beams on both sides: at the first frame whose beam lists differ, is every beam
    that only one side kept within `win` of the worst score the OTHER side kept (i.e. the cut fell between equal scores)?"""
    kw = {k: v for k, v in dkw.items() if k not in ("hotwords", "hotword_weight")}
    okw = dict(dkw)
    kw["hotword_scorer"] = HotwordScorer.build_scorer(dkw["hotwords"], weight=dkw["hotword_weight"])
    beams, c1, c2 = dec.get_starting_state()
    st = orc.get_starting_state()
    x64 = _oracle_input(x)
    T = x.shape[0]
    for t in range(T):
        try:
            beams = dec.partial_decode_beams(x[t:t + 1], c1, c2, beams, t, is_end=(t == T - 1), **kw)
            with np.errstate(all="ignore"):
                ob = orc.partial_decode_beams(x64[t:t + 1], st, t, is_end=(t == T - 1), **okw)
        except ValueError:
            return False
        g = {(b.text, b.partial_word): b.lm_score for b in beams}
        e = {(o.text, o.partial): o.lm for o in ob}
        if list(g) == list(e):
            continue
        if set(g) == set(e):  # same beams in another order: equal scores swapped
            return all(abs(g[k] - e[k]) <= win for k in g) and _order_is_tie(list(g), list(e), e, win)
        if not g or not e:
            return False
        g_cut, e_cut = min(g.values()), min(e.values())
        only_g = [v for k, v in g.items() if k not in e]
        only_e = [v for k, v in e.items() if k not in g]
        return all(abs(v - e_cut) <= win for v in only_g) and all(abs(v - g_cut) <= win for v in only_e)
    return False


def _order_is_tie(a, b, score, win):
    return all(x == y or abs(score[x] - score[y]) <= win for x, y in zip(a, b))


def run_case(rng, execute=True):
    dec, orc, x, dkw = one_case(rng)
    if not execute:  # replaying the generator up to the case of interest (FUZZ_ONLY)
        if x.shape[0] >= 2:
            if rng.random() < 0.5:
                rng.integers(0, x.shape[0] + 1, size=2)
        return "skipped"
    if os.environ.get("FUZZ_TRACE"):
        print("   V=%d T=%d dtype=%s lm=%s %r" % (x.shape[1], x.shape[0], x.dtype, type(dec._language_model).__name__, dkw), flush=True)
    x64 = _oracle_input(x)
    with np.errstate(all="ignore"):
        try:
            exp = orc.decode_beams(x64, sniff_on=x, **dkw)
            err = None
        except ValueError as e:
            exp, err = None, e
    try:
        got = dec.decode_beams(x, **dkw)
    except ValueError:
        assert err is not None, "product raised ValueError, oracle did not"
        return "both raise"
    assert err is None, "oracle raised %r, product did not" % (err,)
    tol = TOL
    # float32 rows: exact in the default mode (the product restates the reference's float32 log-softmax; the oracle is fed the
    # float32 matrix itself) -- except PROBABILITY input, whose scores the reference accumulates in float32 altogether
    with np.errstate(all="ignore"):
        f32_prob = x.dtype == np.float32 and x.shape[0] > 0 and math.isclose(x.sum(axis=1).mean(), 1)
    f32_poly = x.dtype == np.float32 and _prune_mode() == "p" and x.shape[1] <= 4095
    f32_path = TOL_F32 is not None and (f32_poly or (x.dtype == np.float16 and x.shape[1] % 8 == 0 and x.shape[1] <= 1024))
    tkw = {"tol": TOL_F32, "tie_tol": 4e-5} if f32_path else {"tol": TOL, "tie_tol": 1e-9}
    if f32_prob:  # (every backend: the reference's float32 score accumulation is not restated)
        tkw = {"tol": 1e-4, "tie_tol": 4e-5}
    expd = [{"text": e[0], "frames": [[w, int(a), int(b)] for w, (a, b) in e[2]], "logit": e[3], "lm": e[4]} for e in exp]
    try:
        check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in got], expd, what="whole", **tkw)
    except AssertionError:
        # A near-tie (scores equal to ~1 ulp: quantised fp16 / integer logits) that straddles a cut -- beam width,
        # score threshold, history prune -- is decided by the last bit of exp/log and legitimately differs between
        # libm and numpy; from there on the two searches hold different beams. Recognised by: every beam that only
        # one side returned has a twin on the other side with the same score to 1e-9.
        gm = {o.text: o.lm_score for o in got}
        em = {e["text"]: e["lm"] for e in expd}
        only_g = [v for t, v in gm.items() if t not in em]
        only_e = [v for t, v in em.items() if t not in gm]
        win = tkw["tie_tol"]
        common_ok = all(abs(gm[t] - em[t]) <= max(win, tkw["tol"]) for t in gm if t in em)
        twins = lambda a, b: all(any(abs(v - w) <= win for w in b) for v in a)  # noqa: E731
        if common_ok and only_g and only_e and twins(only_g, list(em.values())) and twins(only_e, list(gm.values())):
            return "near-tie at a cut"
        # ... or sits EXACTLY on the score threshold (quantised logits make score differences exact multiples of the
        # quantum, e.g. best - 3.0 with beam_prune_logp = -3.0): `>=` holds on one side and fails by one ulp on the other
        thr = max(list(gm.values()) + list(em.values())) + dkw["beam_prune_logp"]
        on_thr = lambda a: all(abs(v - thr) <= win for v in a)  # noqa: E731
        if common_ok and (only_g or only_e) and on_thr(only_g) and on_thr(only_e):
            return "near-tie at a cut"
        # ... or the tie was at a cut in the MIDDLE of the utterance and the two searches have long since gone separate
        # ways (no twins at the end): step both sides frame by frame and look at the first frame they part at
        if _parted_at_a_tie(dec, orc, x, dkw, win):
            return "near-tie at a cut"
        raise
    # batch entry points: ragged batch (incl. an empty utterance) == utterance by utterance
    if rng.random() < 0.25:
        T0 = x.shape[0]
        parts = [x, x[: T0 // 2], x[:0], x[T0 // 3:]]
        bkw = {k: v for k, v in dkw.items() if k != "prune_history"}
        texts = dec.decode_batch(None, parts, **bkw)
        singles = [dec.decode(p_, **bkw) for p_ in parts]
        assert texts == singles, ("decode_batch != decode", texts, singles)
        bb = dec.decode_beams_batch(None, parts, **dkw)
        for p_, beams_b in zip(parts, bb):
            one = dec.decode_beams(p_, **dkw)
            assert [(o.text, o.text_frames, o.logit_score, o.lm_score) for o in one] == [
                (o.text, o.text_frames, o.logit_score, o.lm_score) for o in beams_b], "decode_beams_batch != decode_beams"
    # stateful decoding: the second half from the first half's last LM state (tests/test_decoder.py:426-456)
    if dec._language_model is not None and x.shape[0] >= 4 and rng.random() < 0.25:
        h = x.shape[0] // 2
        first = dec.decode_beams(x[:h], **dkw)
        with np.errstate(all="ignore"):
            ofirst = orc.decode_beams(x64[:h], sniff_on=x[:h], **dkw)
        if first and ofirst and first[0].text == ofirst[0][0]:
            second = dec.decode_beams(x[h:], lm_start_state=first[0].last_lm_state, **dkw)
            with np.errstate(all="ignore"):
                osecond = orc.decode_beams(x64[h:], lm_start_state=ofirst[0][1], sniff_on=x[h:], **dkw)
            expd2 = [{"text": e[0], "frames": [[w, int(a), int(b)] for w, (a, b) in e[2]], "logit": e[3], "lm": e[4]} for e in osecond]
            try:
                check_beams([(o.text, o.text_frames, o.logit_score, o.lm_score) for o in second], expd2, what="stateful", **tkw)
            except AssertionError:
                if x.dtype == np.float16:  # quantised logits: near-ties at cuts (see above)
                    return "near-tie at a cut"
                raise
    # the same utterance in chunks through partial_decode_beams
    T = x.shape[0]
    if T >= 2 and rng.random() < 0.5:
        cuts = sorted(set([0, T] + [int(c) for c in rng.integers(0, T + 1, size=2)]))
        kw = {k: v for k, v in dkw.items() if k not in ("hotwords", "hotword_weight")}
        kw["hotword_scorer"] = HotwordScorer.build_scorer(dkw["hotwords"], weight=dkw["hotword_weight"])
        beams, c1, c2 = dec.get_starting_state()
        st = orc.get_starting_state()
        okw = {k: v for k, v in dkw.items()}
        for a, b in zip(cuts[:-1], cuts[1:]):
            beams = dec.partial_decode_beams(x[a:b], c1, c2, beams, a, is_end=(b == T), **kw)
            # chunked vs chunked: the BPE force_next_break flag is local to a call (decoder.py:442), so a chunk
            # boundary may legitimately change the result -- the oracle is cut at the same places
            with np.errstate(all="ignore"):
                ob = orc.partial_decode_beams(x64[a:b], st, a, is_end=(b == T), sniff_on=x[a:b], **okw)
            gotc = [(bm.text + "|" + bm.partial_word, [(str(k), f) for k, f in enumerate(bm.text_frames)] + [("p", bm.partial_frames)],
                     bm.logit_score, bm.lm_score) for bm in beams]
            expc = [{"text": o.text + "|" + o.partial, "frames": [[str(k), int(f[0]), int(f[1])] for k, f in enumerate(o.tframes)]
                     + [["p", int(o.pframes[0]), int(o.pframes[1])]], "logit": o.logit, "lm": o.lm} for o in ob]
            try:
                check_beams(gotc, expc, what="chunk %d:%d" % (a, b), **tkw)
            except AssertionError:
                if x.dtype == np.float16:  # quantised logits: near-ties at the beam-width cut (see above)
                    return "near-tie at a cut"
                raise
        return "ok+chunked"
    return "ok"


def run_many(n, seed, tol=None, tol_f32=None):
    """n random cases from `seed` against whatever library pyctcdecode_amd currently uses; returns the outcome counts.
    tol_f32: bound for float32 inputs of a multiple of four labels on the HIP build (its packed float32 exponential)."""
    global TOL, TOL_F32
    if tol is not None:
        TOL = tol
    TOL_F32 = tol_f32
    rng = np.random.default_rng(seed)
    stats = {}
    for _ in range(n):
        r = run_case(rng)
        stats[r] = stats.get(r, 0) + 1
    return stats


def main():
    warnings.simplefilter("ignore")
    logging.disable(logging.CRITICAL)
    use_backend()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    stats = {}
    for i in range(n):
        state = rng.bit_generator.state
        if os.environ.get("FUZZ_TRACE"):
            print("case", i, flush=True)
        only = os.environ.get("FUZZ_ONLY")
        try:
            first = int(os.environ.get("FUZZ_FROM", "0"))  # FUZZ_FROM=k: replay the generator up to case k, execute from there
            r = run_case(rng, execute=(i >= first) if only is None else int(only) == i)
        except Exception:
            print("FAILED case %d (seed %d); rng state before the case:\n%r" % (i, seed, state))
            raise
        stats[r] = stats.get(r, 0) + 1
    print("%s == oracle on %d random cases (seed %d): %s" % ("hip" if HIP else "sim", n, seed, stats))


if __name__ == "__main__":
    main()
