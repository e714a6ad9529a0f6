"""GPU: `python bench.py --gpus 2` as a PLAIN command starts two ranks itself (torch.distributed.run, rendezvous on 127.0.0.1)
and rank 0 prints one line with n_gpus == 2 and the training leg's gradient-exchange block filled.  On the 1-GPU test box the
two ranks share the device and exchange over gloo (the hooks bench.py documents); on an 8-GPU node the same command runs one
rank per GPU over RCCL.  Reference launch line: train_stage1.sh:11 (`torchrun --nproc_per_node=4`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_command_with_gpus_2_launches_two_ranks():
    env = dict(os.environ, G4R_DIST_BACKEND="gloo", G4R_FORCE_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--llama-layers", "2",
           "--batch", "2", "--no-cpu-baseline", "--no-roofline", "--no-extras", "--decode-tokens", "0", "--train-steps", "1",
           "--train-batch", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo" and line["scaling"] == "weak"
    assert line["value"] > 0 and line["config"]["valid"] is False          # 2 decoder layers: a launch test, not a measurement
    ex = line["train"]["exchange"]
    assert ex is not None and ex["overlapped_with_backward"] and ex["buckets"]


def test_gpus_flag_and_launcher_world_size_must_agree():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
