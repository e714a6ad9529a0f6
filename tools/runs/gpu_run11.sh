#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2j; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemv or attn_decode" -x > $O/pytest_dec.log 2>&1
echo "rc $?" >> $O/pytest_dec.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_dec.log | tail -12 | cut -c1-260
timeout 600 python tools/decode_bench.py --tokens 64 > $O/decode.log 2>&1; tail -2 $O/decode.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_decode -o dec --output-format csv -- python $GRAFT_REPO_ROOT/tools/decode_bench.py --tokens 32 > $O/prof_decode.log 2>&1 )
f=$(ls $O/prof_decode/*kernel_stats.csv $O/prof_decode/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/decode_kernel_stats.csv; rm -rf $O/prof_decode
head -12 $O/decode_kernel_stats.csv | cut -c1-230
timeout 1200 python -m pytest tests/test_pipeline_gpu.py tests/test_generation_gpu.py -q -m gpu > $O/pytest_pipe.log 2>&1
echo "rc $?" >> $O/pytest_pipe.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_pipe.log | tail -12 | cut -c1-260
