#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2m; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probe/gemm_bench
timeout 300 $B --rounds 11 g:4616,4096,1024,24 g:4616,4096,1024,0 g:4616,4096,1024,8 g:4616,4096,1024,1 g:4616,3328,1024,24 g:4616,768,1024,0 g:4616,768,1024,7 g:4616,768,1024,10 \
  g:4616,1024,1024,0 g:4616,1024,1024,7 g:4616,1024,1024,6 g:4616,1024,1024,10 g:4616,1024,1024,11 g:4616,1024,1024,15 \
  g:4616,1024,4096,0 g:4616,1024,4096,7 g:4616,1024,4096,6 g:4616,1024,4096,24,3 g:4616,1024,4096,24,2 g:4616,1024,4096,8 \
  g:4616,3072,1024,24 g:4616,3072,1024,0 > $O/vit_b8_gemms.jsonl 2> $O/vit_b8_gemms.err
python - <<'PY'
import json, os
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2m/vit_b8_gemms.jsonl"):
    d = json.loads(l); print(d["case"], d["median_us"], d["TFLOPs_median"], "bad", d["checked_bad"])
PY
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "greedy or argmax" > $O/pytest_g.log 2>&1; tail -2 $O/pytest_g.log
timeout 600 python tools/decode_bench.py --tokens 64 > $O/decode.log 2>&1; tail -1 $O/decode.log
timeout 600 python tools/decode_bench.py --tokens 64 --sample > $O/decode_s.log 2>&1; tail -1 $O/decode_s.log
