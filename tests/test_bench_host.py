"""Host-side logic of bench.py that needs no GPU: the CPU-baseline legs (oracle workers as child processes), the
self-launch of N ranks, the roofline bookkeeping."""

import json
import os
import subprocess
import sys

import pytest
from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_cpu_baseline_legs_run_the_oracle_in_child_processes():
    res = bench.cpu_baseline(3, 16, 8, 0.3)
    assert res["kind"] == "port" and res["cores"] == 1 and res["unit"] == "grid-point-updates/s"
    assert res["value"] > 0 and "16^3" in res["sample"] and "one thread" in res["sample"]
    allc = res["all_cores"]
    # ONE problem of the single-thread leg's size on all cores (the OpenMP build of the same C source)
    assert 1 <= allc["cores"] <= 512 and allc["value"] > 0 and "ONE Poisson 3-D 16^3" in allc["sample"]


def test_cpu_bench_worker_prints_one_json_line():
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "2", "32", "0.2"], cwd=ROOT, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["cells"] == 32 * 32 and d["epochs"] >= 1 and d["seconds"] > 0


def test_plain_gpus_flag_starts_one_rank_per_gpu_as_child_processes(monkeypatch):
    """`python bench.py --gpus 4` outside a launcher: torch.distributed.run with 4 ranks on 127.0.0.1, this
    script and its own arguments -- and nothing else happens in the parent."""
    seen = dict()

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(key, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the children's status is the parent's
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.basename(cmd[-5]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_roofline_object_reports_physical_and_model_fractions():
    r = bench.roofline("k", model_bytes=11e9, moved_bytes=8e9, ms=2.0, traffic=9e9, traffic_source="profiles/x.json")
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["achieved"] - 4500.0) < 1e-9 and abs(r["frac"] - 4500.0 / 8000.0) < 1e-12  # counter bytes / time
    assert abs(r["achieved_model"] - 5500.0) < 1e-9 and abs(r["frac_model"] - 5500.0 / 8000.0) < 1e-12
    r = bench.roofline("k", 11e9, 8e9, 2.0, None, None)
    assert r["traffic"] is None and abs(r["achieved"] - 4000.0) < 1e-9 and "compulsory" in r["traffic_source"]
    per_update, S = bench.algorithmic_bytes_per_update(3, 9, 8)
    assert abs(per_update - 131.4285707) < 1e-4 and abs(S - 8 / 7) < 1e-7
