#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "pyramid_levels or conv3x3" -x > $O/pytest_conv.log 2>&1
echo "rc $?" >> $O/pytest_conv.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_conv.log | tail -8 | cut -c1-260
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_train_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -k "spi or region or end_to_end or fuse or stage1 or config" > $O/pytest_pipe.log 2>&1
echo "rc $?" >> $O/pytest_pipe.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_pipe.log | tail -8 | cut -c1-260
timeout 900 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2p/bench.log").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "single_stream"):
    print(k, json.dumps(d.get(k))[:600])
print({k: v for k, v in d["roofline"].items() if k in ("achieved", "frac", "conv")})
for k, v in list(d["kernels"].items())[:10]: print(k, v)
PY
