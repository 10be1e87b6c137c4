#!/bin/bash
# Round-6 (GPU box, repo root): the shader clock during beam_wave (GRBM_GUI_ACTIVE per XCD / duration of the dispatch) after the
# exact float32 prune stage and after the faster polynomial one (CTCDEC_PRUNE_EXP=pk) -- is the beam stage slower behind a faster
# prune stage because the chip clocks lower?
set -u
export TMPDIR=/tmp
out=gpurun_out/${OUT:-r06u}
mkdir -p $out
for mode in np pk np pk; do
  tag=clock_${mode}_$RANDOM
  CTCDEC_PRUNE_EXP=$mode rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/$tag.d -o $tag -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-shard --no-peaky --no-extras > $out/$tag.log 2>&1
  python - "$out/$tag.d" "$mode" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root, mode = sys.argv[1], sys.argv[2]
dur = {}
for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cyc = defaultdict(float)
for p in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Dispatch_Id"]] += float(r["Counter_Value"])
for kern in ("beam_wave", "frame_prune_fast"):
    rows = [(d, dur[d][1], cyc[d]) for d in dur if kern in dur[d][0] and d in cyc]
    rows = rows[2:]  # (warm-up dispatches)
    if rows:
        ns = sum(r[1] for r in rows) / len(rows)
        c = sum(r[2] for r in rows) / len(rows) / 8.0
        print("CLOCK prune=%s %-18s %d dispatches: %.3f ms, %.1f M cycles per XCD -> %.3f GHz" % (mode, kern, len(rows), ns / 1e6, c / 1e6, c / ns))
PY
  rm -rf $out/$tag.d
done
