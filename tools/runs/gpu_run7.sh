#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
B=tools/probe/gemm_bench
timeout 300 $B --rounds 11 g:4096,4096,4096,0 g:4096,4096,4096,7 g:767,4096,4096,7 g:767,4096,4096,0 g:767,4096,11008,7 \
  g:577,3072,1024,14 g:577,1024,1024,14 g:577,4096,1024,13 g:577,1024,4096,14,2 g:4616,3072,1024,0 g:4616,1024,4096,0 \
  c:1,48,48,1024,4,2 c:1,24,24,1024,4,8 g:36864,1024,1088,0 > $O/gemm_epi2.jsonl 2> $O/gemm_epi2.err
cut -c1-200 $O/gemm_epi2.jsonl
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py tests/test_pipeline_gpu.py -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest.log | tail -20 | cut -c1-220
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.log 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2f/bench.log").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "single_stream", "decode", "train"):
    print(k, json.dumps(d.get(k))[:600])
r = d["roofline"]; print("roofline", {k: v for k, v in r.items() if k not in ("vit",)}); print("vit", r.get("vit"))
for k, v in list(d["kernels"].items())[:12]: print(k, v)
PY
