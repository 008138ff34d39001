"""CPU: `python bench.py --gpus N` (the driver's command shape, no launcher around it) starts its N ranks itself, before any GPU call,
and fails loudly - a non-zero exit, never a silent one-GPU run - when it cannot (reference: one process per GPU,
/root/reference/eval_dense.py:29-32)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SR_BENCH_SHARE_GPU")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)


no_gpu = pytest.mark.skipif(torch.cuda.device_count() > 0, reason="written for the CPU container: needs a host without GPUs")


@no_gpu
def test_too_few_devices_is_an_error_not_a_downgrade():
    out = _run(["--gpus", "2", "--steps", "1"])
    assert out.returncode == 2
    assert "needs 2 visible GPUs" in out.stderr and not out.stdout.strip()


def test_launch_command_is_one_rank_per_gpu_on_this_node():
    out = _run(["--gpus", "4", "--steps", "2", "--warmup", "1", "--print-launch"], SR_BENCH_SHARE_GPU="1")
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout)["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "2", "--warmup", "1"]          # the same arguments, minus the dry-run flag


def test_a_launcher_with_another_world_size_is_refused():
    out = _run(["--gpus", "2", "--steps", "1"], WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    assert out.returncode == 2 and "WORLD_SIZE=3" in out.stderr


@no_gpu
def test_failed_ranks_give_a_non_zero_exit():
    # no GPU here: both ranks die at their first CUDA call; the parent must report it, not print a line
    out = _run(["--gpus", "2", "--steps", "1", "--no-cpu-baseline"], SR_BENCH_SHARE_GPU="1")
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "2-rank run exited with status" in out.stderr
