"""`python bench.py --gpus N` started as a plain command (how the driver starts it) spawns its own ranks
(torch.distributed.run, one per GPU) before anything touches a GPU, relays rank 0's JSON line and the child's exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_without_a_gpu_the_ranks_fail_and_the_exit_code_is_relayed():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for hosts without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                          "--batch", "8", "--no-configs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert "torch.distributed" in res.stderr or "ChildFailedError" in res.stderr      # the ranks were started
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_plain_command_with_two_ranks_prints_one_line(tmp_path):
    """Two ranks share this box's one GPU over gloo (the rehearsal backend; RCCL needs a GPU per rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak" and "gloo" in line["backend"]
    assert line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1 and line["gather_us"] > 0
    assert sorted((r["config"], r["dtype"], r["n_gpus"]) for r in line["configs"]) == [(3, "f64", 2), (4, "f32", 2), (4, "f64", 2)]
