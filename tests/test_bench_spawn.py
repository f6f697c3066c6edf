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
    # what ran where: one identity card per rank, the size of the process group as the ranks saw it
    B = line["config"]["batch_per_gpu"]
    assert line["world_seen"] == 2 and [c["rank"] for c in line["ranks"]] == [0, 1]
    assert [c["shard"] for c in line["ranks"]] == [[0, B], [B, 2 * B]]
    assert all(c["name"] and c["pid"] > 0 and "device" in c and "pci_bus_id" in c and c["placement"]["candidates"] >= 1 for c in line["ranks"])
    assert len({c["pid"] for c in line["ranks"]}) == 2
    cfg3 = [r for r in line["configs"] if r["config"] == 3][0]
    assert cfg3["expected_scaling"]["per_gpu_batch"]["8"] == 128
    # the native C++ leg, started by rank 0 as a child once the other rank had gone: one device here (two gloo ranks shared it)
    nm = line["native_multi"]
    assert "error" not in nm, nm
    assert nm["n_gpus"] == 1 and "note_devices" in nm and nm["value"] > 0 and nm["gather_us"] > 0 and "rccl" in nm["rccl"]["library"]
    assert "ranks_still_alive_at_start" not in nm


@pytest.mark.gpu
def test_the_rccl_calls_of_the_multi_rank_path_run_with_one_rank():
    """A box with one GPU cannot hold two RCCL ranks, but a process group of ONE rank on the nccl backend issues the same
    calls as the N > 1 path on the same kinds of tensors: communicator creation on this GPU, barrier(device_ids), the
    all-reduce (MAX) of the timings and warm-up counts, the asynchronous all-gather (buffers in rotation) of the objectives out
    of the buffer the finalizing waves write.  bench.py --single-rank-collectives; the stated configs run sharded as well."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-rank-collectives", "--steps", "5", "--warmup", "2",
                          "--batch", "2048", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and "rccl" in line["backend"] and "ONE rank" in line["backend"]
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1 and line["gather_us"] > 0
    assert any(r.get("gather_us", 0) > 0 for r in line["configs"] if r["mode"] == "device-resident" and r["config"] in (3, 4))
    # what RCCL saw
    assert line["world_seen"] == 1 and len(line["ranks"]) == 1
    card = line["ranks"][0]
    assert card["rank"] == 0 and card["device"] == 0 and card["shard"] == [0, 2048] and card["name"] and card["pci_bus_id"]
    assert line["rccl"]["torch_nccl_version"]
    # the native C++ host path in the same line: the same step -- launch + gather -- through tolfg_multi over
    # device 0 and real RCCL, the step loop issued from native code
    nm = line["native_multi"]
    assert "error" not in nm, nm
    assert nm["n_gpus"] == 1 and nm["issue"] == "grouped" and nm["batch_per_gpu"] == 2048 and nm["steps"] == 5
    assert nm["value"] > 0 and nm["eval_us"] > 0 and nm["gather_us"] > 0 and nm["ms_per_step"] >= 1e-3 * nm["eval_us"] * 0.98
    assert "rccl" in nm["rccl"]["library"] and nm["rccl"]["version_code"] > 0
    assert nm["devices"][0]["pci_bus_id"] == card["pci_bus_id"]
    # the same work per step on the same device (5 steps only, and the two processes' output buffers land in different placement
    # classes: a loose bound; the driver's line compares the two at 20 steps on 8192 trajectories)
    assert 0.5 < nm["ms_per_step"] / line["ms_per_step"] < 2.0
    assert nm["objectives_stored_to_host_instead"]["ms_per_step"] > 0
    assert sorted((r["config"], r["dtype"]) for r in nm["configs"]) == [(3, "f64"), (4, "f32"), (4, "f64")]


def test_a_process_group_of_another_size_than_gpus_is_refused():
    """`--gpus 1` under a launcher that started two ranks: no line under the wrong n_gpus (exit code 2, before any GPU work)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2"], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 2 and "WORLD_SIZE" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_what_makes_a_line_unreportable():
    """bench.rank_problems: the checks behind exit code 4, on made-up identity cards."""
    sys.path.insert(0, ROOT)
    import bench
    card = lambda r, bus, host="n0": {"rank": r, "device": r, "pci_bus_id": bus, "host": host}      # noqa: E731
    good = [card(r, "0000:%02x:00.0" % (0x10 + r)) for r in range(8)]
    assert bench.rank_problems(good, 8, 8, "nccl") == []
    assert any("--gpus says 4" in p for p in bench.rank_problems(good, 8, 4, "nccl"))
    twice = good[:7] + [card(7, good[0]["pci_bus_id"])]
    assert any("pairwise distinct" in p for p in bench.rank_problems(twice, 8, 8, "nccl"))
    assert bench.rank_problems(twice, 8, 8, "gloo") == []                 # a gloo rehearsal may share GPUs
    other_host = good[:7] + [card(7, good[0]["pci_bus_id"], host="n1")]   # the same bus id on another host is another device
    assert bench.rank_problems(other_host, 8, 8, "nccl") == []
    assert any("identity cards" in p for p in bench.rank_problems(good[:7], 8, 8, "nccl"))
    no_bus = [dict(card(r, None)) for r in range(2)]                      # no bus id known: the ordinals tell them apart
    assert bench.rank_problems(no_bus, 2, 2, "nccl") == []
    assert bench.rank_problems([card(0, None)], 1, 1, None) == []


def test_rank_zero_waits_for_the_other_ranks_to_be_gone():
    sys.path.insert(0, ROOT)
    import bench
    child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(0.4)"])
    left = bench.wait_for_pids([os.getpid(), child.pid], 5.0)           # its own pid is not waited for
    assert left == [] and child.poll() is not None
    sleeper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(5)"])
    try:
        assert bench.wait_for_pids([sleeper.pid], 0.2) == [sleeper.pid]  # bounded: it says who is still there
    finally:
        sleeper.kill()
        sleeper.wait()


def test_the_native_leg_without_a_gpu_yields_a_record_that_says_so():
    """The native C++ leg is a child process: whatever happens in it, the parent gets a record (and prints its own line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for hosts without a GPU")
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    args = argparse.Namespace(steps=2, warmup=1, batch=8, ts=20, mission="S10", aircraft="tempest", dtype="f64", x_buffers=1, global_batch=0,
                              no_configs=True)
    rec = bench.run_native_child(1, args, "grouped", timeout=300)
    assert "error" in rec and "visible" in rec["error"] and rec["child_exit_code"] == 1 and rec["n_gpus"] == 1
