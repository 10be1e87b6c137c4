"""Summarise rocprofv3 --pmc output (counter_collection.csv files under a directory): per kernel and counter
the mean value per dispatch.  Usage: python tools/pmc_summary.py <dir> [out.json]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("ctc::be::", "").replace("void ", "")
    return name


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                per_dispatch[(k, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[k] = {"grid": row.get("Grid_Size"), "wg": row.get("Workgroup_Size"), "vgpr": row.get("VGPR_Count"),
                           "sgpr": row.get("SGPR_Count"), "lds": row.get("LDS_Block_Size"), "scratch": row.get("Scratch_Size")}
        for (k, _d), ctrs in per_dispatch.items():
            for c, v in ctrs.items():
                acc[k][c].append(v)
    out = {}
    for k, ctrs in acc.items():
        out[k] = {"dispatches": max(len(v) for v in ctrs.values()), **meta.get(k, {})}
        for c, v in sorted(ctrs.items()):
            out[k][c] = sum(v) / len(v)
    txt = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
