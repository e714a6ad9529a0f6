#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2g; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probe/gemm_bench
timeout 300 $B --rounds 9 c:1,192,192,1024,24,1 c:1,96,96,1024,24,1 c:1,48,48,1024,24,2 c:1,48,48,1024,4,2 c:1,24,24,1024,4,8 \
  c:1,24,24,1024,14,4 g:36864,1024,9216,24 > $O/conv_tapinner.jsonl 2> $O/conv_tapinner.err
cut -c1-220 $O/conv_tapinner.jsonl
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d $O/pmc_conv -o conv -- $B --rounds 3 c:1,192,192,1024,24,1 c:1,96,96,1024,24,1 > $O/pmc_conv.log 2>&1 )
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r2g"
acc = collections.defaultdict(list)
for f in glob.glob(O + "/pmc_conv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:60], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n", len(v), "mean", sum(v) / len(v))
PY
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_autograd_gpu.py -q -m gpu -x > $O/pytest_train.log 2>&1
echo "rc $?" >> $O/pytest_train.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_train.log | tail -12 | cut -c1-220
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv or fuse" > $O/pytest_conv.log 2>&1
echo "rc $?" >> $O/pytest_conv.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest_conv.log | tail -8 | cut -c1-220
