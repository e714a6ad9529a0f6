#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probe/gemm_bench
timeout 300 $B --rounds 9 g:6272,1024,36864,1 g:6272,1024,36864,0 g:6272,1024,36864,8 g:6272,1024,36864,24,2 g:6272,1024,36864,24,3 g:6272,1024,36864,24,5 \
  g:6272,1024,36864,26,5 g:6272,1024,36864,26,2 g:6272,1024,36864,7 g:6272,1024,36864,0,2 g:6272,1024,36864,1,2 \
  g:767,4096,4096,7 g:767,4096,4096,24,4 g:767,4096,4096,26,4 g:767,4096,4096,24,5 g:767,4096,4096,0 g:767,4096,4096,6 > $O/pconv.jsonl 2> $O/pconv.err
python - <<'PY'
import json, os
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2q/pconv.jsonl"):
    d = json.loads(l); print(d["case"], d["median_us"], d["TFLOPs_median"], "bad", d["checked_bad"])
PY
