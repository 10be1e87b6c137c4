"""Register / LDS / spill table of every kernel in libctcdec.so: recompiles the three device translation units with the flags
pyctcdecode_amd/build.py uses plus -Rpass-analysis=kernel-resource-usage and prints one line per kernel.
  python tools/kernel_resources.py [name-filter] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyctcdecode_amd import build  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = []
    srcs = ("backend_hip.hip", "beam_wave_hip.hip", "beam_group_hip.hip")
    from concurrent.futures import ThreadPoolExecutor

    def remarks(src):
        path = os.path.join(ROOT, "pyctcdecode_amd", "csrc", src)
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-Wno-unused-result"]
        cmd = [build.hipcc()] + flags + build.HIP_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", path,
                                                                        "-o", "/dev/null"]
        return subprocess.run(cmd, capture_output=True, text=True).stderr

    with ThreadPoolExecutor(3) as ex:
        errs = list(ex.map(remarks, srcs))
    for src, err in zip(srcs, errs):
        cur = None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    dm = demangle([r["name"] for r in rows])
    print("%-70s %5s %5s %5s %8s %10s %10s %4s %8s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "sgpr-spill", "vgpr-spill", "occ", "LDS"))
    for r in rows:
        name = dm.get(r["name"], r["name"])
        if flt and flt not in name:
            continue
        print("%-70s %5s %5s %5s %8s %10s %10s %4s %8s" % (
            name[:70], r.get("TotalSGPRs", r.get("SGPRs", "?")), r.get("VGPRs", "?"), r.get("AGPRs", "?"),
            r.get("ScratchSize [bytes/lane]", "?"), r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"),
            r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))


if __name__ == "__main__":
    main()
