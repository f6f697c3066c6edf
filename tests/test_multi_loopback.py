"""The native several-GPUs-one-process host path (tol_amd/csrc/multi.cpp) with MORE THAN ONE part, on a one-GPU box.

RCCL refuses a device list that names a device twice, so the parts share device 0 through the test seam
TOLFG_MULTI_SHARED_DEVICES=1 and the collectives go through tests/loopback_nccl (the eight nccl* entry points multi.cpp
resolves, implemented with device-to-device copies), selected with TOLFG_RCCL_LIBRARY.  What this pins: the issuing
threads, the shard dealing (13/12/12; a part with an EMPTY shard), the per-shard wind-table offsets, the padded gather and
its re-ordering, the all-reduce of the partial sums.  What it does not: RCCL's transport (tests/test_multi_native.py calls
the real library on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_close

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LOOP = os.path.join(HERE, "loopback_nccl")


def loopback_library():
    subprocess.run(["make", "-s", "-C", LOOP], check=True)
    return os.path.join(LOOP, "libloopback_nccl.so")


def test_loopback_library_exports_what_multi_resolves():
    import ctypes as C
    lib = loopback_library()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    with open(os.path.join(ROOT, "tol_amd", "csrc", "multi.cpp")) as fh:
        src = fh.read()
    import re
    want = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(want) == 8 and want <= have
    assert C  # (the library itself needs a HIP runtime in the process to load: done by the GPU tests)


def test_an_explicit_collective_library_that_cannot_be_loaded_is_an_error_not_a_second_pick(tolfg, tmp_path):
    """TOLFG_RCCL_LIBRARY is final: with a path that does not load, creation fails and says so (fresh process: the
    choice is made once).  Runs without a GPU too -- the message then is the missing device, checked on the GPU box."""
    code = ("import tol_amd as t\n"
            "try:\n"
            "    t.Multi('S10', ['tempest'], ts=20, devices=[0])\n"
            "except t.TolfgError as e:\n"
            "    print('ERR', e.code, e)\n")
    env = dict(os.environ, TOLFG_RCCL_LIBRARY=str(tmp_path / "no_such_library.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and "ERR" in out.stdout, out.stdout + out.stderr
    import torch
    if torch.cuda.is_available():
        assert "TOLFG_RCCL_LIBRARY" in out.stdout and "cannot be loaded" in out.stdout


def run_worker(tmp_path, mission, dtype, total, parts, N, wind):
    out = str(tmp_path / f"multi_{mission}_{dtype}_{total}_{parts}_{wind.replace(':', '_')}.npz")
    env = dict(os.environ, TOLFG_MULTI_SHARED_DEVICES="1", TOLFG_RCCL_LIBRARY=loopback_library(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(HERE, "multi_worker.py"), out, mission, dtype, str(total), str(parts), str(N), wind],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("mission,dtype,total,parts,N,wind", [
    ("S10", "f64", 37, 3, 100, "shear"),       # 13 / 12 / 12
    ("S10", "f64", 37, 3, 64, "table"),        # per-shard wind-table offsets
    ("mixed", "f64", 23, 4, 200, "shear"),     # 6 / 6 / 6 / 5, both missions in every shard
    ("G7", "f32", 7, 4, 52, "shear"),          # 2 / 2 / 2 / 1
    ("S10", "f64", 2, 3, 40, "shear"),         # fewer trajectories than parts: part 2 holds nothing
])
def test_several_parts_on_one_device(tolfg, oracle, tmp_path, mission, dtype, total, parts, N, wind):
    from tol_amd.distributed import shard_bounds
    import multi_worker as W
    r = run_worker(tmp_path, mission, dtype, total, parts, N, wind)
    assert "loopback_nccl" in str(r["library"])
    shards = [tuple(int(v) for v in s) for s in r["shards"]]
    assert shards == [shard_bounds(total, i, parts) for i in range(parts)]
    if (total, parts) == (37, 3):
        assert [hi - lo for lo, hi in shards] == [13, 12, 12]
    n, neF, neG = (int(v) for v in r["sizes"])
    F1, G1 = r["Fs"], r["Gs"]      # the single batch
    # every shard's F and G equal the rows of the single batch, bit for bit
    for i, (lo, hi) in enumerate(shards):
        if hi == lo:
            assert f"F{i}" not in r
            continue
        for name, got, want in (("X", r[f"X{i}"][:, :n], r["Xs"][lo:hi]), ("F", r[f"F{i}"][:, :neF], F1[lo:hi]), ("G", r[f"G{i}"][:, :neG], G1[lo:hi])):
            bad = np.argwhere(got != want)
            assert bad.size == 0, (f"{name} of shard {i} [{lo},{hi}): {len(bad)} of {got.size} entries differ, first at {bad[:6].tolist()}: "
                                   f"{[(got[tuple(b)], want[tuple(b)]) for b in bad[:6]]}")
    # gathered objectives: global order, equal to the single batch's F[:, 0]; the second gather too
    assert r["obj"].shape == (total,)
    assert np.array_equal(r["obj"], F1[:, 0]) and np.array_equal(r["obj_again"], F1[:, 0]) and np.array_equal(r["obj_host"], F1[:, 0])
    assert float(r["mean_host"]) == pytest.approx(float(F1[:, 0].astype(np.float64).mean()), rel=1e-12 if dtype == "f64" else 1e-6)
    assert float(r["mean"]) == pytest.approx(float(F1[:, 0].astype(np.float64).mean()), rel=1e-12 if dtype == "f64" else 1e-6)
    # and against the oracle
    if dtype == "f64":
        trajs = W.trajectories(tolfg, mission, total)
        air = ["tempest", "skywalker"]
        for t in sorted({0, total - 1, shards[1][0], shards[-1][0] if shards[-1][1] > shards[-1][0] else 0}):
            tr = trajs[t]
            kw = dict(N=N, radius_goal=tr.radius_goal, start=(tr.xi, tr.yi, tr.zi))
            if wind == "table":
                o = oracle.Problem(tr.mission, air[tr.aircraft], wind_table=r["tables"][t], **kw)
            else:
                o = oracle.Problem(tr.mission, air[tr.aircraft], Vref=tr.Vref, href=tr.href, **kw)
            Fo, Go = o.eval(r["Xs"][t][:o.n])
            assert_close(F1[t][:len(Fo)], Fo, what=f"F of trajectory {t}")
            assert_close(G1[t][:len(Go)], Go, mask=o.undefined_mask(), what=f"G of trajectory {t}")


@pytest.mark.gpu
@pytest.mark.parametrize("issue,gather", [("grouped", "rccl"), ("threads", "rccl"), ("grouped", "host"), ("threads", "host")])
@pytest.mark.parametrize("mission,dtype,total,parts,N", [("mixed", "f64", 23, 4, 200), ("S10", "f32", 9, 4, 52)])
def test_asynchronous_gather_equals_the_synchronous_one(tolfg, tmp_path, mission, dtype, total, parts, N, issue, gather):
    """tolfg_multi_step / gather_begin / gather_wait on 4 loop-back parts -- 11 steps on different inputs over
    4 rotating objective buffers, the host waiting for gather j-1 with evaluation j already issued -- deliver bitwise what
    eval_from + the synchronous gather_objectives deliver, in both ways of issuing the collective (one group call; one call
    per device thread) and both ways of gathering (ncclAllGather into device vectors; item 7: the finalizing waves storing
    straight into one pinned host vector, no collective).  A ticket expires after 4 further gathers; the native step loop
    (tolfg_multi_time_steps) leaves the object in working order; switching the way of gathering changes no number."""
    import json
    r = run_worker(tmp_path, mission, dtype, total, parts, N, f"pipeline:{issue}:{gather}")
    assert "loopback_nccl" in str(r["library"])
    a, s = r["obj_async"], r["obj_sync"]
    assert a.shape == s.shape == (11, total) and np.isfinite(a).all()
    assert np.array_equal(a, s)
    assert list(r["tickets"]) == list(range(11))
    assert not np.array_equal(a[0], a[1])                                  # the inputs really differ from step to step
    assert np.array_equal(a[0], r["obj_single"])                           # set 0 = the initial guesses: the single batch's objectives
    assert str(r["expired"]).startswith(str(tolfg.capi.ERR_ARG)) and "ticket" in str(r["expired"])
    assert np.array_equal(r["obj_run_last"], s[int(r["run_last_set"])])
    tol = 1e-12 if dtype == "f64" else 1e-6
    assert float(r["mean_last"]) == pytest.approx(float(s[-1].astype(np.float64).mean()), rel=tol)
    assert int(r["soak_mismatches"]) == 0                                  # 400 pipelined steps, the host three gathers behind
    assert np.array_equal(r["obj_after_loop"], s[1])
    if "refusal" in r:           # one device's launch refused: the error comes back, nobody hangs, the next step is right
        assert str(r["refusal"]).startswith(str(tolfg.capi.ERR_HIP)) and "lost an objective partial" in str(r["refusal"]) and "device" in str(r["refusal"])
        assert np.array_equal(r["obj_after_refusal"], s[0])
    assert np.array_equal(r["obj_other_gather"], s[2])
    assert float(r["mean_other_gather"]) == pytest.approx(float(s[2].astype(np.float64).mean()), rel=tol)
    tim = json.loads(str(r["timing"]))
    for t, with_gather in zip(tim, (True, False, True)):
        assert t["issue"] == issue and t["gather"] == gather and t["devices"] == parts
        assert t["wall_us_per_step"] > 0 and t["launch_us_per_step"] > 0 and len(t["launch_us_per_device"]) == parts
        assert (t["gather_us"] > 0) == with_gather
