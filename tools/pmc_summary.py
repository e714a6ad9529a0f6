#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files:  pmc_summary.py DIR [DIR ...] -> JSON."""
import csv, glob, json, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name") or r.get("Kernel-Name") or ""
            c = r.get("Counter_Name") or ""
            v = float(r.get("Counter_Value") or 0)
            a = acc[k][c]
            a[0] += v
            a[1] += 1
out = {}
for k, cs in acc.items():
    if "gemm" in k or "roi_align" in k or "attn" in k:
        out[k[:120]] = {c: dict(mean=a[0] / a[1], n=a[1]) for c, a in cs.items()}
print(json.dumps(out, indent=1))
