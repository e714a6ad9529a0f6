#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2i; mkdir -p $O
timeout 600 python tools/decode_bench.py --tokens 64 > $O/decode.log 2>&1; tail -2 $O/decode.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_decode -o dec --output-format csv -- python $GRAFT_REPO_ROOT/tools/decode_bench.py --tokens 32 > $O/prof_decode.log 2>&1 )
f=$(ls $O/prof_decode/*kernel_stats.csv $O/prof_decode/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/decode_kernel_stats.csv; rm -rf $O/prof_decode
head -25 $O/decode_kernel_stats.csv | cut -c1-200
timeout 900 python -m pytest tests/test_generation_gpu.py -q -m gpu > $O/pytest_gen.log 2>&1
echo "rc $?" >> $O/pytest_gen.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_gen.log | tail -12 | cut -c1-220
