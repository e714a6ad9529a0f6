#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
timeout 1800 python -m pytest tests/test_autograd_gpu.py tests/test_train_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "autograd or train or roi_align" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^FAILED|^E  " $O/pytest.log | tail -30 | cut -c1-220
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.log 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r2d/bench.log").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "single_stream", "roofline", "cpu_baseline", "decode", "train"):
        v = d.get(k)
        if k == "roofline" and v: v = {kk: vv for kk, vv in v.items()}
        print(k, json.dumps(v)[:900])
    for k, v in list(d["kernels"].items())[:14]: print(k, v)
except Exception as e:
    print("bench parse failed", e)
PY
# rocprof stats + counter passes of the serial, eager bench (kernel-trace only)
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --train-steps 0 --decode-tokens 0 --no-roofline"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_stats --output-format csv -- $BENCH ) > $O/prof_stats.log 2>&1
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_mfma --output-format csv -- $BENCH ) > $O/pmc_mfma.log 2>&1
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_fetch --output-format csv -- $BENCH ) > $O/pmc_fetch.log 2>&1
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_write --output-format csv -- $BENCH ) > $O/pmc_write.log 2>&1
python tools/pmc_report.py $O/pmc_report.json $O/pmc_mfma $O/pmc_fetch $O/pmc_write > $O/pmc_report.txt 2>&1
head -30 $O/pmc_report.txt
# keep only the small csv summaries (the raw counter csvs are large)
find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
