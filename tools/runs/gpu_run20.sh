#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2t; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probe/gemm_bench
cases=""
for shape in 12288,4096 4096,4096 22016,4096 4096,11008; do for t in 4 12 13 14 0; do cases="$cases g:8,$shape,$t"; done; done
cases="$cases g:8,4096,11008,14,4 g:8,4096,11008,13,4 g:8,4096,4096,14,2 g:8,4096,4096,14,4 g:16,12288,4096,13 g:16,22016,4096,13 g:32,22016,4096,13"
timeout 300 $B --rounds 11 $cases > $O/smallm.jsonl 2> $O/smallm.err
python - <<'PY'
import json, os
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2t/smallm.jsonl"):
    d = json.loads(l); gb = 2.0 * d["N"] * d["K"] / 1e9
    print(d["case"], d["median_us"], "GB/s", round(gb / d["median_us"] * 1e6), "bad", d["checked_bad"])
PY
