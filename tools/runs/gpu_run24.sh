#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
timeout 900 python -m pytest tests/test_fullwidth_gpu.py tests/test_pipeline_gpu.py -q -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -2 $O/pytest.log
timeout 600 python bench.py --train-steps 0 > $O/bench.log 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2y/bench.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["single_stream"], d["decode"]["ms_per_token"], d["decode"].get("batched"), d["cpu_baseline"]["value"])
PY
