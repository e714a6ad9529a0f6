"""CPU: the host logic bench.py gained in round 5 -- `--gpus N` as a plain command re-launches itself under
torch.distributed.run (the launcher of the reference's own scripts, train_stage1.sh:11), the deadline around the multi-rank
training leg, and the dispatch rule / argument validation of the batched decode projection (include/g4r_kernels.h)."""
import ctypes
import os
import sys
import time
from types import SimpleNamespace

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    import bench as b
    return b


def test_plain_gpus_n_builds_the_launcher_command(bench, monkeypatch):
    seen = {}
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("G4R_FORCE_DEVICE", raising=False)
    assert bench.self_launch(SimpleNamespace(gpus=4)) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_plain_gpus_n_refuses_more_ranks_than_devices(bench, monkeypatch):
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("G4R_FORCE_DEVICE", raising=False)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.self_launch(SimpleNamespace(gpus=8))


def test_training_leg_deadline(bench):
    assert bench.with_deadline(lambda: 41 + 1, 5.0, 0) == 42
    with pytest.raises(ZeroDivisionError):
        bench.with_deadline(lambda: 1 / 0, 5.0, 0)
    t0 = time.time()
    with pytest.raises(bench._Deadline):
        bench.with_deadline(lambda: time.sleep(30), 0.3, 0)          # a rank stuck in a collective: give up, keep the line
    assert time.time() - t0 < 5


def test_batched_projection_dispatch_rule_and_abi_validation():
    from gpt4roi_amd import _lib
    from gpt4roi_amd import kernels as K
    # 2..4 rows, one staging pass, K <= 8192 (profiles/r05_gemv_batch.txt); 8 requests (config 5) stay on the small-M tiles
    assert K.gemv_batch_wins(2, 12288, 4096) and K.gemv_batch_wins(4, 22016, 4096) and K.gemv_batch_wins(3, 32006, 4096)
    assert not K.gemv_batch_wins(8, 12288, 4096) and not K.gemv_batch_wins(1, 4096, 4096)
    assert not K.gemv_batch_wins(4, 4096, 11008) and not K.gemv_batch_wins(2, 4096, 4000)          # long K; K % 64
    lib = _lib.lib()
    f = lib.g4r_gemv_batch_bf16
    f.restype = ctypes.c_int
    P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    f.argtypes = [P, I, L, P, ctypes.c_float, P, P, L, P, P, L, I, I, I, I, I, I, P]
    assert f(None, 1, 4096, None, 1e-6, None, None, 4096, None, None, 0, 4096, 4096, 4096, 0, 0, 0, None) == 1
    assert b"2 <= B <= 16" in lib.g4r_last_error()
    assert f(None, 4, 4096, None, 1e-6, None, None, 4096, None, None, 0, 4096, 4000, 4096, 0, 0, 0, None) == 1
    assert b"multiple of 64" in lib.g4r_last_error()
    assert f(None, 4, 4096, None, 1e-6, None, None, 4096, None, None, 0, 4096, 4096, 4096, 0, 0, 0, None) == 1
    assert b"null pointer" in lib.g4r_last_error()


def test_forward_flops_per_request_match_the_survey(bench):
    """config.target_arithmetic of the line: SURVEY.md 8d puts one request at ~15.8 TFLOP (P = 24, 32 RoIs, T ~ 780) and ~9.4 TFLOP at
    the reference-native P = 16 (T ~ 510); 5000 region-tokens/s at 336^2 is then ~2.47 PF/s = 0.99 of the dense 16-bit peak"""
    f24 = bench.forward_flops_per_request(24, 32, 767)
    f16 = bench.forward_flops_per_request(16, 32, 510)          # (the survey's T ~ 510; the bench's 224^2 prompt is 447 tokens: 8.7 TFLOP)
    assert 15.5e12 < f24 < 16.2e12 and 9.0e12 < f16 < 9.9e12
    assert abs(5000 / 32 * f24 / 1e15 - 2.48) < 0.03
    # the parts SURVEY.md 8d lists: ViT 365 GF (23 layers, P = 24), fuse convs 4 620 GF, pconvs 473 GF at 32 RoIs
    S, C = 577, 1024
    assert abs(23 * (24.0 * S * C * C + 4.0 * S * S * C) / 1e9 - 365) < 2
    assert abs(5 * 2.0 * 9 * C * C * 85 * 576 / 1e9 - 4620) < 5


def test_greedy_record_merger(tmp_path, monkeypatch):
    """tools/merge_greedy.py: the per-(dtype, seed) records of the single-request test and the merged-16 record become ONE file with
    both scopes, which bench.py quotes with the file's hash"""
    import importlib.util
    import json
    root = tmp_path / "repo"
    (root / "tools").mkdir(parents=True)
    (root / "gpurun_out" / "greedy_parity").mkdir(parents=True)
    (root / "profiles").mkdir()
    src = open(os.path.join(ROOT, "tools", "merge_greedy.py")).read()
    (root / "tools" / "merge_greedy.py").write_text(src)
    for dt, seed in (("fp16", 82), ("bf16", 82)):
        json.dump({"dtype": dt, "seed": seed, "new_tokens": 64, "teacher_forced_identical": 64, "free_running_exact_len": 64,
                   "whole_path_generate_exact_len": 64, "whole_path_first_divergence": None, "against": "HF"},
                  open(root / "gpurun_out" / "greedy_parity" / f"{dt}_{seed}.json", "w"))
    json.dump({"dtype": "fp16", "scope": "merged16", "requests": 16, "new_tokens": 64, "merged_exact_len": [64] * 15 + [40]},
              open(root / "gpurun_out" / "greedy_parity" / "merged16_fp16.json", "w"))
    monkeypatch.setattr(sys, "argv", ["merge_greedy.py", "07"])
    spec = importlib.util.spec_from_file_location("merge_greedy_under_test", str(root / "tools" / "merge_greedy.py"))
    spec.loader.exec_module(importlib.util.module_from_spec(spec))
    doc = json.load(open(root / "profiles" / "r07_greedy_parity.json"))
    assert [r["dtype"] for r in doc["runs"]] == ["bf16", "fp16"] and doc["merged16"]["merged_exact_len"][-1] == 40
