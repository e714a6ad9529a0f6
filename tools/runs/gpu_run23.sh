#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2x; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o dec --output-format csv -- python $GRAFT_REPO_ROOT/tools/decode_bench.py --tokens 16 --batch 8 > $O/prof.log 2>&1 )
f=$(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/decode_b8_kernel_stats.csv; rm -rf $O/prof
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2x/decode_b8_kernel_stats.csv")))
for r in rows[:16]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {r["Calls"]:>6} calls {float(r["AverageNs"])/1e3:8.1f} us {r["Percentage"]:>6}%  {r["Name"][:100]}')
PY
