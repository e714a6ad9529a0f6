#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 3000 python -m pytest tests/test_generation_gpu.py tests/test_autograd_gpu.py tests/test_fullwidth_gpu.py tests/test_pipeline_gpu.py tests/test_train_gpu.py -q -m gpu -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |FAILED|Error" $O/pytest.log | tail -40
