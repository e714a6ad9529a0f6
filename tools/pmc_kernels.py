#!/usr/bin/env python3
"""tools/pmc_kernels.py DIR [DIR ...] [--min-us 50] -- per kernel name, the average of EVERY counter found in the rocprofv3
--pmc pass directories (one pass per directory), with launch count and average duration; derived: clock = GRBM_GUI_ACTIVE / 8 /
duration, mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

dirs = [d for d in sys.argv[1:] if not d.startswith("--")]
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 50.0
rep = defaultdict(dict)
for d in dirs:
    by, dur = defaultdict(lambda: defaultdict(float)), {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Kernel_Name"])
            by[key][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: defaultdict(float))
    for (did, name), cs in by.items():
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:70]
        agg[short]["n"] += 1
        agg[short]["us"] += dur[(did, name)]
        for c, v in cs.items():
            agg[short][c] += v
    for k, a in agg.items():
        n = a.pop("n")
        us = a.pop("us") / n
        if us < min_us:
            continue
        rep[k].setdefault("launches", int(n))
        rep[k].setdefault("avg_us", []).append(round(us, 1))
        for c, v in a.items():
            rep[k][c] = v / n
        if "GRBM_GUI_ACTIVE" in a:
            gui = a["GRBM_GUI_ACTIVE"] / n
            rep[k]["clock_GHz"] = round(gui / 8 / us / 1e3, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
                rep[k]["mfma_util"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (gui / 8 * 1024), 4)
for k, v in rep.items():
    print(json.dumps({"kernel": k, **{c: (round(x, 1) if isinstance(x, float) else x) for c, x in v.items()}}))
