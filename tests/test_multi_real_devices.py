"""The multi-GPU paths on a node that HAS several MI355X -- marker `multigpu`, never part of `-m gpu` (the build sessions' boxes and the
driver's test box have one GPU: THESE TESTS HAVE NOT RUN YET; they are the first thing to run on such a node:
`python -m pytest tests/test_multi_real_devices.py -m multigpu -x -q`).

They are the one-GPU tests with the seams taken out: the workers of tests/test_multi_loopback.py with one REAL device per part and the
real RCCL (no TOLFG_MULTI_SHARED_DEVICES, no TOLFG_RCCL_LIBRARY), and bench.py started the way the driver starts it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.multigpu


def devices_here():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def run_worker(tmp_path, mission, dtype, total, parts, N, what):
    out = str(tmp_path / f"real_{mission}_{dtype}_{total}_{parts}_{what.replace(':', '_')}.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("TOLFG_MULTI_SHARED_DEVICES", "TOLFG_RCCL_LIBRARY")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    res = subprocess.run([sys.executable, os.path.join(HERE, "multi_worker.py"), out, mission, dtype, str(total), str(parts), str(N), what],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("wind", ["shear", "table"])
def test_shards_on_real_devices_equal_the_single_batch(tolfg, tmp_path, wind):
    """Every device's F and G bitwise the rows of the single batch on device 0; objectives all-gathered by RCCL over xGMI in global order."""
    n = devices_here()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    parts, total, N = min(n, 8), 37, 100
    from tol_amd.distributed import shard_bounds
    r = run_worker(tmp_path, "mixed" if wind == "shear" else "S10", "f64", total, parts, N, wind)
    assert "rccl" in str(r["library"]) and "loopback" not in str(r["library"])
    shards = [tuple(int(v) for v in s) for s in r["shards"]]
    assert shards == [shard_bounds(total, i, parts) for i in range(parts)]
    nn, neF, neG = (int(v) for v in r["sizes"])
    for i, (lo, hi) in enumerate(shards):
        if hi > lo:
            assert np.array_equal(r[f"F{i}"][:, :neF], r["Fs"][lo:hi], equal_nan=True) and np.array_equal(r[f"G{i}"][:, :neG], r["Gs"][lo:hi], equal_nan=True), i
    F0 = r["Fs"][:, 0]
    assert np.array_equal(r["obj"], F0) and np.array_equal(r["obj_again"], F0) and np.array_equal(r["obj_host"], F0)
    assert float(r["mean"]) == pytest.approx(float(F0.mean()), rel=1e-12)


@pytest.mark.parametrize("issue,gather", [("grouped", "rccl"), ("threads", "rccl"), ("grouped", "host"), ("threads", "host")])
def test_pipelined_steps_on_real_devices(tolfg, tmp_path, issue, gather):
    """The asynchronous gather over real devices: 11 + 400 pipelined steps bitwise the synchronous gather, in both ways of issuing the
    collective and both ways of gathering; a refused launch on one device leaves nobody waiting."""
    n = devices_here()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    parts, total = min(n, 8), 61
    r = run_worker(tmp_path, "mixed", "f64", total, parts, 200, f"pipeline:{issue}:{gather}")
    a, s = r["obj_async"], r["obj_sync"]
    assert a.shape == s.shape == (11, total) and np.array_equal(a, s) and np.array_equal(a[0], r["obj_single"])
    assert int(r["soak_mismatches"]) == 0
    if parts >= 3 and "refusal" in r:
        assert str(r["refusal"]).startswith(str(tolfg.capi.ERR_HIP)) and np.array_equal(r["obj_after_refusal"], s[0])
    assert np.array_equal(r["obj_other_gather"], s[2])
    for t in json.loads(str(r["timing"])):
        assert t["devices"] == parts and t["issue"] == issue and t["gather"] == gather and all(v > 0 for v in t["launch_us_per_device"])


def test_bench_line_over_all_devices(tmp_path):
    """bench.py as the driver starts it at N > 1: one line; N identity cards on N different devices; the stated configs; the native
    C++ leg over the same devices (per-thread issue) and the bracket form beside it."""
    n = devices_here()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    n = min(n, 8)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["world_seen"] == n and "rccl" in line["backend"] and line["scaling"] == "weak"
    assert len({c["pci_bus_id"] for c in line["ranks"]}) == n and [c["rank"] for c in line["ranks"]] == list(range(n))
    assert line["value"] > 0 and line["gather_us"] > 0
    nm = line["native_multi"]
    assert "error" not in nm, nm
    assert nm["n_gpus"] == n and nm["issue"] == "threads" and len(nm["eval_us_per_device"]) == n and nm["value"] > 0
    assert "error" not in line["native_multi_grouped"], line["native_multi_grouped"]
    # weak scaling: the job's rate is not far from N times a rank's own launch rate
    assert line["value"] > 0.5 * n * 8192 * 200 / (line["roofline"]["kernel_ms"] * 1e-3)
