#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2l; mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --streams 1 --no-graph --no-cpu-baseline --train-steps 3 --decode-tokens 0 --no-roofline"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_train -o tr --output-format csv -- $BENCH ) > $O/prof_train.log 2>&1
f=$(ls $O/prof_train/*kernel_stats.csv $O/prof_train/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/train_kernel_stats.csv; rm -rf $O/prof_train
python - <<'PY'
import csv, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2l"
rows = list(csv.DictReader(open(O + "/train_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {r["Calls"]:>6} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]:>6}%  {r["Name"][:110]}')
PY
tail -2 $O/prof_train.log | cut -c1-600
