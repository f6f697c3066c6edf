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
