#!/usr/bin/env python3
"""Per-kernel report from rocprofv3 counter passes (one directory per --pmc pass, each with --kernel-trace):

    pmc_report.py OUT.json DIR_mfma [DIR_fetch [DIR_write]]

DIR_mfma: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   -> effective clock and MFMA pipe utilisation per kernel
          util = MFMA_BUSY / (GUI_ACTIVE/8 XCDs x 1024 SIMDs);  clock = GUI_ACTIVE / 8 / duration
DIR_fetch / DIR_write: --pmc FETCH_SIZE / --pmc WRITE_SIZE -> HBM bytes per launch (read = 2 x FETCH_SIZE x 1024: the
          gfx950 correction of MI355X_MICROARCH.md section HBM; write = WRITE_SIZE x 1024, uncalibrated)
Averages are over every dispatch of a kernel name in the pass."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    by = defaultdict(lambda: defaultdict(float))
    meta = {}
    for r in rows:
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        by[key][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: defaultdict(float))
    for (did, name), cs in by.items():
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:80]
        a = agg[short]
        a["n"] += 1
        a["us"] += meta[(did, name)]
        for c, v in cs.items():
            a[c] += v
    return agg


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    rep = {}
    mf = load(dirs[0]) if dirs else {}
    for k, a in mf.items():
        n = a["n"]
        gui = a.get("GRBM_GUI_ACTIVE", 0.0) / n
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
        us = a["us"] / n
        rep[k] = {"launches": int(n), "avg_us": round(us, 2), "clock_GHz": round(gui / 8 / us / 1e3, 3) if us else None,
                  "mfma_util": round(busy / (gui / 8 * 1024), 4) if gui else None, "mfma_busy_cycles": busy}
    for idx, (cname, key, mul) in enumerate((("FETCH_SIZE", "hbm_read_bytes", 2048.0), ("WRITE_SIZE", "hbm_write_bytes", 1024.0))):
        if len(dirs) > idx + 1:
            for k, a in load(dirs[idx + 1]).items():
                rep.setdefault(k, {"launches": int(a["n"]), "avg_us": round(a["us"] / a["n"], 2)})[key] = int(a.get(cname, 0.0) / a["n"] * mul)
    rep = dict(sorted(rep.items(), key=lambda kv: -(kv[1].get("avg_us", 0) * kv[1].get("launches", 0))))
    # which kernel sources these counters belong to: bench.py compares this with the tree it runs from and says so on its line
    # (a stale file must not pass for a fresh measurement)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    csrc = os.path.join(root, "gpt4roi_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    rep["_meta"] = {"kernel_sources_sha256_16": h.hexdigest()[:16], "launches": 0}
    json.dump(rep, open(out_path, "w"), indent=1)
    for k, v in list(rep.items())[:25]:
        if k == "_meta":
            continue
        print(f"{k[:60]:60s} n={v['launches']:5d} {v.get('avg_us', 0):9.1f}us clk={v.get('clock_GHz')} util={v.get('mfma_util')} "
              f"rd={v.get('hbm_read_bytes')} wr={v.get('hbm_write_bytes')}")


if __name__ == "__main__":
    main()
