"""bench.py's N > 1 entry: `python bench.py --gpus N` must get as far as the GPU on any box -- it launches itself under
torch.distributed.run when the driver has not, defaults to BASELINE config 5, and names the digest it will hold the stitched
stream to.  No GPU here: the launch is checked as a plan, and once for real up to the point where a rank asks for its device."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    a = types.SimpleNamespace(gpus=2, virtual=False, dry_run=False, single_process=False)
    a.__dict__.update(kw)
    return a


def test_launch_plan_one_rank_per_gpu_over_rccl():
    plan = bench.launch_plan(_args(gpus=8), ["--gpus", "8", "--steps", "5"], n_dev=8)
    assert "error" not in plan
    cmd = plan["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert "MI355_BENCH_BACKEND" not in plan["env"]  # RCCL (backend "nccl") is the default
    assert plan["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launch_plan_fewer_devices_than_ranks():
    plan = bench.launch_plan(_args(gpus=4), ["--gpus", "4"], n_dev=1)
    assert "error" in plan and "--virtual" in plan["error"]
    plan = bench.launch_plan(_args(gpus=4, virtual=True), ["--gpus", "4", "--virtual", "--dry-run"], n_dev=1)
    assert plan["env"]["MI355_BENCH_BACKEND"] == "gloo" and "--dry-run" not in plan["cmd"]
    # (torch.distributed.run's parser reads --virtual as an abbreviation of its own --virtual-local-rank: it travels in the environment)
    assert "--virtual" not in plan["cmd"] and plan["env"]["MI355_BENCH_VIRTUAL"] == "1"


def test_default_workload_for_several_gpus_is_config5_with_committed_digests():
    """1 GiB a rank of ONE N GiB web-text input: 8 GiB at N = 8 (BASELINE config 5); the oracle's digests of the totals the
    driver's 2 / 4 / 8 GPU runs produce are committed"""
    for n in (2, 4, 8):
        g = bench.committed_digest("webtext", n << 30, "default")
        assert g is not None and g["out_len"] > 0 and len(g["out_sha256"]) == 64, n
    assert bench.committed_digest("webtext", 12345, "default") is None
    assert bench.committed_digest("enwik8", 8 << 30, "default") is None


def test_dry_run_prints_the_launch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--virtual", "--dry-run"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert plan["env"]["MI355_BENCH_BACKEND"] == "gloo" and "torch.distributed.run" in plan["cmd"]


@pytest.mark.timeout(300)
def test_self_launch_reaches_the_device_check():
    """the real thing on a box without a GPU: both ranks start under torch.distributed.run, make their part of the input and stop
    where the product needs its device (there is no CPU path to fall back to)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU box runs the bench itself")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--virtual", "--size", "1048576", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode != 0
    assert "bench.py needs a GPU" in out.stderr, out.stderr[-2000:]
    assert "dry run: 2 ranks" in out.stderr
