"""Turn gpurun_out/r02/ (written by tools/r02_profile.sh on the GPU box) into the tracked summaries under profiles/."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02")
DST = os.path.join(ROOT, "profiles")
KIB = 1024.0


def load(name):
    with open(os.path.join(SRC, name)) as f:
        return json.load(f)


def stats(path):
    with open(path) as f:
        rows = list(csv.DictReader(f))
    return "\n".join("%-70s calls %4s  avg %12.1f us  total %6.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                        float(r["Percentage"])) for r in rows)


def main():
    out = {
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- "
                "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shard --no-peaky [--batch 512]  (tools/r02_profile.sh, "
                "tools/pmc_run.sh); counter values are KiB per dispatch, mean over the 3 dispatches of a pass; 4096 (and 512) "
                "utterances x T=1000 x V=1024, beam 100, 4-gram + hot words",
        "correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced "
                      "streaming read (16 B/lane) -> doubled for frame_prune_f32x4 (float4 loads; the doubled value lands within "
                      "0.01 % of the algorithmic T*V*4 bytes, which is the calibration). The beam kernels do scattered 16-128 B "
                      "accesses: the guide calls those widths uncalibrated, so their FETCH_SIZE is reported raw "
                      "(fetch_bytes_corrected == raw). WRITE_SIZE is uncalibrated everywhere (raw).",
    }
    for b in (4096, 512):
        f, w = load("fetch_%d.json" % b), load("write_%d.json" % b)
        d = {}
        for k in f:
            if k.startswith("__amd") or k == "utt_sniff":
                continue
            name = "frame_prune" if k.startswith("frame_prune") else k.split("<")[0]
            raw = f[k]["FETCH_SIZE"] * KIB
            d[name] = {"kernel": k, "dispatches": f[k]["dispatches"], "grid": f[k]["grid"], "FETCH_SIZE_KiB_raw": f[k]["FETCH_SIZE"],
                       "WRITE_SIZE_KiB_raw": w[k]["WRITE_SIZE"], "fetch_bytes_corrected": raw * 2 if name == "frame_prune" else raw,
                       "write_bytes_raw": w[k]["WRITE_SIZE"] * KIB, "algorithmic_bytes": b * 1000 * 1024 * 4.0}
        out["batch_%d" % b] = d
    with open(os.path.join(DST, "r02_pmc_hbm_traffic.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    sq, sq2 = load("sq1_4096.json"), load("sq2_4096.json")
    merged = {k: {**sq[k], **sq2.get(k, {})} for k in sq if not k.startswith("__amd")}
    with open(os.path.join(DST, "r02_pmc_sq.json"), "w") as fo:
        json.dump({"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> on python bench.py --steps 2 --warmup 1 "
                           "--no-cpu-baseline --no-shard --no-peaky (batch 4096); mean per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                           "SQ_ACTIVE_INST_* are quad-cycles summed over waves", "kernels": merged}, fo, indent=1, sort_keys=True)
    with open(os.path.join(DST, "r02_kernel_stats.txt"), "w") as fo:
        fo.write("rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                 "--no-shard --no-peaky   (default batch: 4096 utterances x T=1000 x V=1024 on one MI355X)\n"
                 + stats(os.path.join(SRC, "kernel_stats_4096.csv")) + "\n\n"
                 "same with --batch 512 (the round-1 shard; the launcher picks the workgroup kernel when utterances <= 2 x CUs)\n"
                 + stats(os.path.join(SRC, "kernel_stats_512.csv")) + "\n")
    for a, b in (("bench.json", "r02_bench.json"), ("bench.log", "r02_bench.log"), ("pytest_gpu.log", "r02_pytest_gpu.log"),
                 ("bench_dist1.json", "r02_bench_forced_dist_1gpu.json")):
        shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
    print(open(os.path.join(DST, "r02_kernel_stats.txt")).read())
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.startswith(("fetch", "write"))} for k, v in out["batch_4096"].items()}, indent=1))
    per = merged.get("beam_wave<104>", {})
    if per:
        w = per["SQ_WAVES"] * 1000.0
        print("beam_wave per wave-frame: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f quad-cycles %.0f wait %.0f" % (
            per["SQ_INSTS_VALU"] / w, per["SQ_INSTS_SALU"] / w, per["SQ_INSTS_LDS"] / w, per["SQ_INSTS_VMEM_RD"] / w,
            per["SQ_INSTS_VMEM_WR"] / w, per["SQ_WAVE_CYCLES"] / w, per["SQ_WAIT_ANY"] / w))
    pr = merged.get("frame_prune_f32x4<4>", {})
    if pr:
        print("frame_prune per row: VALU %.0f SALU %.0f LDS %.0f" % (pr["SQ_INSTS_VALU"] / pr["SQ_WAVES"], pr["SQ_INSTS_SALU"] / pr["SQ_WAVES"],
                                                                     pr["SQ_INSTS_LDS"] / pr["SQ_WAVES"]))


if __name__ == "__main__":
    main()
