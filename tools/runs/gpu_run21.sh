#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -q -m gpu -k "attn_decode or batched or greedy or decode or gemm_plain or gemv" -x > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest.log | tail -10 | cut -c1-260
for b in 4 8 16; do timeout 600 python tools/decode_bench.py --tokens 32 --batch $b 2>&1 | tail -1; done | tee $O/decode_batch.log
