"""RCCL itself under the slab paths' communication layer (`odil_amd.slab.TorchDistComm`) on a single-GPU box: a process
group of one rank that is its own neighbour (tools/rccl_selfloop.py, run as a fresh process)."""
import os
import subprocess
import sys

import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_comm_layer_over_rccl_with_the_rank_as_its_own_neighbour():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selfloop.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "rccl self-loop ok" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
