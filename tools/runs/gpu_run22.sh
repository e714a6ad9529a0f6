#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2v; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_decode --output-format csv -- python $GRAFT_REPO_ROOT/tools/decode_bench.py --tokens 8 > $O/pmc_decode.log 2>&1 )
python - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2v"
acc = collections.defaultdict(list)
for f in glob.glob(O + "/pmc_fetch_decode/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemv" in n or "attn_decode" in n or "greedy" in n:
            acc[(n.split("(")[0][-60:], r["Grid_Size"])].append(float(r["Counter_Value"]))
out = open(O + "/decode_fetch.txt", "w")
for k, v in sorted(acc.items()):
    line = f"{k[0]:62s} grid {k[1]:>8s} n={len(v):5d} FETCH_SIZE mean {sum(v)/len(v):10.1f} KiB  -> x2 = {2*sum(v)/len(v)*1024/1e6:8.2f} MB read per launch"
    print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_fetch_decode
