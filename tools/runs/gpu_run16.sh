#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
G4R_DIST_BACKEND=gloo G4R_FORCE_DEVICE=0 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 4 --warmup 2 --train-steps 2 --train-batch 4 --decode-tokens 8 > $O/bench2.log 2> $O/bench2.err
echo "rc $?" >> $O/bench2.err
tail -5 $O/bench2.err | cut -c1-400
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r2o/bench2.log").read().strip().splitlines()[-1])
    for k in ("value", "n_gpus", "ms_per_step", "single_stream", "decode", "train"):
        print(k, json.dumps(d.get(k))[:900])
except Exception as e:
    print("parse failed", e)
PY
