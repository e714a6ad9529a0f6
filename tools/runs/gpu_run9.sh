#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2h; mkdir -p $O
B=$GRAFT_REPO_ROOT/tools/probe/gemm_bench
for v in 0 7; do for ctr in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_${ctr}_$v -o conv --output-format csv -- $B --rounds 2 c:1,192,192,1024,24,1,$v c:1,96,96,1024,24,1,$v > $O/pmc_${ctr}_$v.log 2>&1 )
done; done
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r2h"
out = open(O + "/pmc_conv_order.txt", "w")
for d in sorted(glob.glob(O + "/pmc_*_[07]")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:48], r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        line = f"{os.path.basename(d)} {k} n={len(v)} mean={sum(v) / len(v):.1f}"
        print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_*_[07]
