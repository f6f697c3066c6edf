"""The N > 1 path with the HIP kernels: two FRESH ranks (torch.distributed.run, gloo) share the one GPU of
the test box, each evaluates its shard of a 37-trajectory batch with the product library, the objectives
are all-gathered in global order and checked against the oracle.  (RCCL needs one GPU per rank; the
sharding, the per-rank evaluation and the gather order are what this test pins.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_close

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.gpu
@pytest.mark.parametrize("total", [37])
def test_two_ranks_evaluate_shards_on_the_gpu_and_gather(tolfg, oracle, tmp_path, total):
    N = 60
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_worker.py"), out, str(total), str(N)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    got = np.load(out)["objectives"]
    assert got.shape == (total,)
    sys.path.insert(0, HERE)
    import dist_worker as W
    want = np.empty(total)
    for t in range(total):
        tr = W.trajectory(tolfg, t)
        o = oracle.Problem("S10", ("tempest", "skywalker")[tr.aircraft], N=N, radius_goal=100.0, start=(tr.xi, tr.yi, tr.zi),
                           Vref=tr.Vref, href=tr.href)
        want[t] = o.eval(W.decision_vector(o.x0(), t), needG=False)[0][0]
    assert_close(got, want, what="gathered objectives (global order)")


@pytest.mark.gpu
def test_bench_line_carries_the_stated_configs_on_more_than_one_rank(tmp_path):
    """bench.py under torch.distributed.run with two ranks sharing this box's GPU over gloo (the rehearsal backend): the one
    JSON line must hold the weak-scaling headline AND configs[3] / configs[4] as stated (global batches of 1024 and 8192,
    sharded over the ranks), each with the per-rank launch time, the gather time and a fraction of N x 8 TB/s."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--batch", "512",
           "--backend", "gloo", "--no-native-multi"]      # (the native leg: tests/test_bench_spawn.py)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "gloo" in line["backend"]
    assert line["config"]["mission"] == "mixed" and line["config"]["batch_per_gpu"] == 512 and line["config"]["global_batch"] == 1024
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["gather_us"] > 0
    recs = line["configs"]
    got = sorted((r["config"], r["batch"], r["dtype"], r["batch_per_gpu"], r["n_gpus"]) for r in recs)
    assert got == [(3, 1024, "f64", 512, 2), (4, 8192, "f32", 4096, 2), (4, 8192, "f64", 4096, 2)]
    for r in recs:
        assert r["ms_per_step"] > 0 and r["eval_us"] > 0 and r["gather_us"] > 0 and 0 < r["frac_of_hbm_peak"] < 1
        assert "strong scaling" in r["workload"]
