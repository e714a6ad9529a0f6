#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2r; mkdir -p $O
timeout 1800 python -m pytest tests/test_pipeline_gpu.py -q -m gpu > $O/pytest_train.log 2>&1
echo "rc $?" >> $O/pytest_train.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_train.log | tail -8 | cut -c1-260
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv" > $O/pytest_conv.log 2>&1; tail -1 $O/pytest_conv.log
timeout 1200 python bench.py --train-steps 0 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2r/bench.log").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "single_stream", "train"):
    print(k, json.dumps(d.get(k))[:700])
print({k: v for k, v in d["roofline"].items() if k in ("achieved", "frac", "conv")})
for k, v in list(d["kernels"].items())[:12]: print(k, v)
PY
