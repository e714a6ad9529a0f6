#!/usr/bin/env python3
"""tools/merge_greedy.py ROUND -- collect the greedy-id records the GPU tests wrote under gpurun_out/greedy_parity/
(tests/test_fullwidth_gpu.py: one file per (dtype, seed) of the single-request test + merged16_fp16.json of the 16-merged-request
test) into profiles/rROUND_greedy_parity.json, which bench.py quotes on its line with its scope and hash."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "06"
src = os.path.join(ROOT, "gpurun_out", "greedy_parity")
runs = []
for f in sorted(glob.glob(os.path.join(src, "*_*.json"))):
    if os.path.basename(f).startswith("merged16"):
        continue
    runs.append(json.load(open(f)))
doc = {"what": "tests/test_fullwidth_gpu.py::test_bench_model_full_depth_32_layers_vs_hf_fp32_on_the_gpu (SURVEY.md 8d config 2: 336^2 image, "
               "32 RoIs, ViT-L/14 + region module + projector + splice + LLaMA-7B 32 x 4096, T = 767, greedy decode of 64 new tokens, ONE "
               f"request at a time) and ::test_sixteen_merged_requests_... (the dispatch bench.py times) on an MI355X, round {int(rnd)}",
       "runs": sorted(runs, key=lambda r: (r["dtype"], r["seed"]))}
m = os.path.join(src, "merged16_fp16.json")
if os.path.exists(m):
    doc["merged16"] = json.load(open(m))
out = os.path.join(ROOT, "profiles", f"r{rnd}_greedy_parity.json")
json.dump(doc, open(out, "w"), indent=1)
print(out, len(runs), "runs", "merged16" in doc)
