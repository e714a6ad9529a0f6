#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2k; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probe/gemm_bench
cases=""
for shape in 12288,4096 22016,4096 32006,4096; do for v in 0 1 2 3 4 5 6 7; do cases="$cases v:$shape,$v"; done; done
for shape in 4096,4096 4096,11008; do for v in 0 1 2 3 4 5 6 7; do cases="$cases g:1,$shape,0,1,0,0,$((100+v))"; done; done
timeout 300 $B --rounds 21 $cases > $O/gemv_variants.jsonl 2> $O/gemv_variants.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2k"
for l in open(O + "/gemv_variants.jsonl"):
    d = json.loads(l)
    gb = 2.0 * d["N"] * d["K"] / 1e9
    print(d["case"], "median_us", d["median_us"], "min_us", d["min_us"], "GB/s", round(gb / d["median_us"] * 1e6, 0), "bad", d["checked_bad"])
PY
tail -3 $O/gemv_variants.err
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemv or attn_decode" -x > $O/pytest_dec.log 2>&1
echo "rc $?" >> $O/pytest_dec.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_dec.log | tail -6 | cut -c1-260
timeout 600 python tools/decode_bench.py --tokens 64 > $O/decode.log 2>&1; tail -1 $O/decode.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_decode -o dec --output-format csv -- python $GRAFT_REPO_ROOT/tools/decode_bench.py --tokens 32 > $O/prof_decode.log 2>&1 )
f=$(ls $O/prof_decode/*kernel_stats.csv $O/prof_decode/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/decode_kernel_stats.csv; rm -rf $O/prof_decode
grep -E "gemv|attn_decode|advance" $O/decode_kernel_stats.csv | cut -c1-60,200-320
