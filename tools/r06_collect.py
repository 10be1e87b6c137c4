"""Turn gpurun_out/r06f/ (written by tools/r06_profile.sh on the GPU box) into the tracked summaries under profiles/r06_*.
The PMC traffic summary carries the SHA-256 of the library that ran (bench.py refuses it for another binary)."""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06f")
DST = os.path.join(ROOT, "profiles")
KIB = 1024.0


def load(name):
    with open(os.path.join(SRC, name)) as f:
        return json.load(f)


def stats(path):
    with open(path) as f:
        rows = list(csv.DictReader(f))
    return "\n".join("%-78s calls %4s  avg %12.1f us  total %6.2f %%" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                        float(r["Percentage"])) for r in rows)


def main():
    out = {
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- "
                "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shard --no-peaky --no-extras  (tools/r06_profile.sh, "
                "tools/pmc_run.sh); counter values are KiB per dispatch, mean over the 3 dispatches of a pass; 4096 "
                "utterances x T=1000 x V=1024, beam 100, 4-gram + hot words",
        "correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced "
                      "streaming read (16 B/lane) -> doubled for frame_prune_fast (float4 loads; the doubled value lands within "
                      "0.01 % of the algorithmic T*V*4 bytes, which is the calibration). The beam kernel and assemble_texts do "
                      "scattered 1-128 B accesses: the guide calls those widths uncalibrated, so their FETCH_SIZE is reported raw "
                      "(fetch_bytes_corrected == raw). WRITE_SIZE is uncalibrated everywhere (raw).",
    }
    f, w = load("fetch_4096.json"), load("write_4096.json")
    d = {}
    wk = {k.split("<")[0]: k for k in w}  # (the two passes are separate runs: match kernels by their plain names)
    for k in f:
        if k.startswith("__amd") or k.startswith("utt_sniff"):
            continue
        name = "frame_prune" if k.startswith("frame_prune_fast") else k.split("<")[0]
        raw = f[k]["FETCH_SIZE"] * KIB
        wv = w[wk[k.split("<")[0]]]
        assert wv["dispatches"] == f[k]["dispatches"] and wv["grid"] == f[k]["grid"], (k, wv, f[k])
        d[name] = {"kernel": k, "dispatches": f[k]["dispatches"], "grid": f[k]["grid"], "FETCH_SIZE_KiB_raw": f[k]["FETCH_SIZE"],
                   "WRITE_SIZE_KiB_raw": wv["WRITE_SIZE"], "fetch_bytes_corrected": raw * 2 if name == "frame_prune" else raw,
                   "write_bytes_raw": wv["WRITE_SIZE"] * KIB, "algorithmic_bytes": 4096 * 1000 * 1024 * 4.0}
    out["batch_4096"] = d
    with open(os.path.join(SRC, "library.sha256")) as fh:
        out["binary_sha16"] = fh.read().split()[0][:16]  # of pyctcdecode_amd/libctcdec.so on the box that measured
    # ... and the hash of what that library was built from (sources, headers, command lines, compiler versions): a rebuild of the same
    # tree under another root is the same kernels in another file (run this script on the tree that was measured)
    import sys

    sys.path.insert(0, ROOT)
    from pyctcdecode_amd import build as _b

    out["source_sha16"] = _b.source_tag()[:16]
    with open(os.path.join(DST, "r06_pmc_hbm_traffic.json"), "w") as fo:
        json.dump(out, fo, indent=1)
    sq, sq2 = load("sq1_4096.json"), load("sq2_4096.json")
    merged = {k: {**sq[k], **sq2.get(k, {})} for k in sq if not k.startswith("__amd")}
    with open(os.path.join(DST, "r06_pmc_sq.json"), "w") as fo:
        json.dump({"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> on python bench.py --steps 2 --warmup 1 "
                           "--no-cpu-baseline --no-shard --no-peaky --no-extras (batch 4096); mean per dispatch; SQ_WAVE_CYCLES / "
                           "SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves", "kernels": merged}, fo, indent=1, sort_keys=True)
    with open(os.path.join(DST, "r06_kernel_stats.txt"), "w") as fo:
        fo.write("rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                 "--no-shard --no-peaky --no-extras   (default batch: 4096 utterances x T=1000 x V=1024 on one MI355X)\n"
                 + stats(os.path.join(SRC, "kernel_stats_4096.csv")) + "\n\n"
                 "same with --batch 512 (the per-GPU shard of configs[3] over 8 GPUs; the launcher picks the workgroup kernel when "
                 "utterances <= 2 x CUs)\n" + stats(os.path.join(SRC, "kernel_stats_512.csv")) + "\n")
    for t in ("ta_4096", "tcp_4096"):
        if os.path.exists(os.path.join(SRC, t + ".json")):
            shutil.copy(os.path.join(SRC, t + ".json"), os.path.join(DST, "r06_pmc_%s.json" % t.split("_")[0]))
    for a, b in (("micro_valu_rates.txt", "r06_micro_valu_rates.txt"), ("micro_mem_latency.txt", "r06_micro_mem_latency.txt"),
                 ("np_div_check.txt", "r06_np_div_check.txt"), ("wave_times.txt", "r06_wave_times.txt"),
                 ("bench.json", "r06_bench.json"), ("bench.log", "r06_bench.log"), ("pytest_gpu.log", "r06_pytest_gpu.log"),
                 ("phases512.log", "r06_phases_512.log"), ("phases4096.log", "r06_phases_4096.log"),
                 ("fuzz_hip.log", "r06_fuzz_hip.log"), ("host_step.log", "r06_host_step.log")):
        if os.path.exists(os.path.join(SRC, a)):
            shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
    print(open(os.path.join(DST, "r06_kernel_stats.txt")).read())
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.startswith(("fetch", "write"))} for k, v in out["batch_4096"].items()}, indent=1))
    per = next((v for k, v in merged.items() if k.startswith("beam_wave")), {})
    if per:
        w_ = per["SQ_WAVES"] * 1000.0
        print("beam_wave per wave-frame: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f quad-cycles %.0f wait %.0f; VALU active / wave "
              "cycles %.3f" % (per["SQ_INSTS_VALU"] / w_, per["SQ_INSTS_SALU"] / w_, per["SQ_INSTS_LDS"] / w_, per["SQ_INSTS_VMEM_RD"] / w_,
                               per["SQ_INSTS_VMEM_WR"] / w_, per["SQ_WAVE_CYCLES"] / w_, per["SQ_WAIT_ANY"] / w_,
                               per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"]))
    pr = next((v for k, v in merged.items() if k.startswith("frame_prune_fast")), {})
    if pr:
        rows = pr["SQ_WAVES"] * 64.0
        print("frame_prune_fast per row: VALU %.0f SALU %.0f LDS %.1f; VALU active / wave cycles %.3f" % (
            pr["SQ_INSTS_VALU"] / rows, pr["SQ_INSTS_SALU"] / rows, pr["SQ_INSTS_LDS"] / rows,
            pr["SQ_ACTIVE_INST_VALU"] / pr["SQ_WAVE_CYCLES"]))
    ls = next((v for k, v in merged.items() if k.startswith("frame_prune_f32x4_listed")), {})
    if ls:
        print("frame_prune_f32x4_listed: %d waves launched per dispatch (rows handed over by the fast kernel are processed by these)" % ls["SQ_WAVES"])


if __name__ == "__main__":
    main()
