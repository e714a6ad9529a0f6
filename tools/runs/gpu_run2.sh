#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 2400 python -m pytest tests/test_fullwidth_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "fullwidth or production or llama_7b or config1 or gemm or conv3x3" -s > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |vs emulate|full width|identical|differ|margin|agreement" $O/pytest.log | tail -30
