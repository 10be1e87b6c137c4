"""Static instruction mix of one kernel of beam_wave_hip.hip, attributed to the inline call path it was compiled from
(the `.loc ... @[ inlined-at ]` chains of -gline-tables-only), as a tree of beam_wave.h functions.
  python tools/isa_by_source.py [kernel-substring] [--reuse] [--depth N] [--under PATH-SUBSTRING] [--lines PATH-SUBSTRING]
Static counts: multiply by how often a path runs per frame (pass1 and what it calls ~1.9x on the bench input) to compare
with the PMC's per-wave-frame numbers."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyctcdecode_amd import build  # noqa: E402

SRC = os.path.join(ROOT, "pyctcdecode_amd", "csrc")
KINDS = ["valu", "vmov", "salu", "wait", "lds", "vmem"]


def functions_of(path):
    """{line: function name} for a header, by brace depth (functions marked CTC_HD / __device__)."""
    out, depth, cur = {}, 0, None
    pat = re.compile(r"(?:CTC_HD|__device__)\s+(?:__forceinline__\s+)?(?:static\s+)?[\w:<>\*&\s]+?\b(\w+)\s*\(")
    for no, line in enumerate(open(path, errors="replace"), 1):
        if cur is None:
            m = pat.search(line)
            if m:
                cur = [m.group(1), depth, False]
        if cur is not None:
            out[no] = cur[0]
        opened = line.count("{")
        depth += opened - line.count("}")
        if cur is not None:
            if opened:
                cur[2] = True
            if cur[2] and depth <= cur[1]:
                cur = None
            elif not cur[2] and ";" in line and depth <= cur[1]:
                cur = None
    return out


def kind_of(op):
    if op.startswith("v_"):
        return "vmov" if op.startswith("v_mov") or op.startswith("v_accvgpr") else "valu"
    if op.startswith("s_"):
        return "wait" if op in ("s_waitcnt", "s_nop") else "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.split("_")[0] in ("global", "flat", "buffer", "scratch"):
        return "vmem"
    return "other"


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    pos = []
    skip = False
    for a in sys.argv[1:]:
        if skip:
            skip = False
        elif a in ("--depth", "--under", "--lines"):
            skip = True
        elif not a.startswith("--"):
            pos.append(a)
    want = pos[0] if pos else "beam_waveILi100ELi4ELb0"
    depth = int(opt("--depth", "3"))
    under = opt("--under")
    lines = opt("--lines")
    os.makedirs("/tmp/isa", exist_ok=True)
    asm = "/tmp/isa/beam_wave_g.s"
    src = os.path.join(SRC, "beam_wave_hip.hip")
    if "--reuse" not in sys.argv or not os.path.exists(asm):
        cmd = [build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-Wno-unused-result"]
        cmd += build.HIP_FLAGS["beam_wave_hip.hip"] + os.environ.get("CTCDEC_HIPCC_EXTRA", "").split()
        cmd += ["-gline-tables-only", "-S", "--cuda-device-only", "-o", asm, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    fmap = {}
    for h in os.listdir(SRC):
        if h.endswith((".h", ".hip")):
            fmap[h] = functions_of(os.path.join(SRC, h))

    def name(f, ln):
        base = os.path.basename(f)
        if base in fmap:
            return fmap[base].get(ln, "%s:%d" % (base, ln))
        return base.split(".")[0]

    loc_re = re.compile(r"(/[^\s:]+):(\d+):\d+")
    inside = False
    path = ("prologue",)
    inner = ""
    agg = collections.defaultdict(collections.Counter)
    by_line = collections.defaultdict(collections.Counter)
    for raw in open(asm):
        if not inside:
            if raw.startswith("_Z") and want in raw.split(":")[0]:
                inside = True
            continue
        if raw.startswith(".Lfunc_end"):
            break
        if raw.lstrip().startswith(".loc"):
            chain = loc_re.findall(raw)
            if chain:
                names = [name(f, int(ln)) for f, ln in chain][::-1]  # outermost first
                dedup = []
                for n in names:
                    if not dedup or dedup[-1] != n:
                        dedup.append(n)
                path = tuple(dedup)
                inner = "%s:%s" % (os.path.basename(chain[0][0]), chain[0][1])
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)\s", raw)
        if not m:
            continue
        k = kind_of(m.group(1))
        full = "/".join(path)
        if under and under not in full:
            continue
        agg["/".join(path[:depth])][k] += 1
        if lines and lines in full:
            by_line[inner][k] += 1
    print("%-60s" % "inline path" + "".join("%7s" % k for k in KINDS))
    tot = collections.Counter()
    for fn, c in sorted(agg.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["vmov"] + kv[1]["salu"])):
        print("%-60s" % fn[-60:] + "".join("%7d" % c[k] for k in KINDS))
        tot.update(c)
    print("%-60s" % "TOTAL" + "".join("%7d" % tot[k] for k in KINDS))
    if lines:
        print("\ninnermost source lines under %s:" % lines)
        for loc, c in sorted(by_line.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["vmov"] + kv[1]["salu"]))[:80]:
            print("%-60s" % loc + "".join("%7d" % c[k] for k in KINDS))


if __name__ == "__main__":
    main()
