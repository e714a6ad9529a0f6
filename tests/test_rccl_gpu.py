"""GPU: the collective calls of the training exchange executed by RCCL (backend "nccl" on ROCm) on device buffers, in a
one-rank process group (a single MI355X is what this environment has; the world-2 semantics run over gloo in
test_grad_reduce_gloo.py / test_sharded_gloo.py / test_train_exchange_gloo.py).  VERDICT r03 missing-3: nothing had ever
proven that RCCL accepts the in-place reduce_scatter_tensor of grad_reduce.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_exchange_collectives_run_on_rccl():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_PORT="29617", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_world1.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
