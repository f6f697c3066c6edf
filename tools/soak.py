#!/usr/bin/env python3
"""Soak: many evaluations back to back on rotating X buffers, every one checked bit for bit against the first evaluation of its X
buffer (64-bit integer checksums of F, G and the objectives, taken on the device after every launch).  The kernel is deterministic
-- fixed summation orders, no floating-point atomics -- so ANY difference is a race (arrival counters, polled partial slots,
workspace reuse between launches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tol_amd
import bench as BN

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
SHAPES = [("S10", "f64", 1024, "reference"), ("mixed", "f64", 8192, "reference"), ("mixed", "f32", 2048, "reference"), ("S10", "f64", 4096, "compact"),
          ("G7", "f64", 37, "reference"), ("mixed", "f32", 8192, "reference"), ("S10", "f64", 1, "reference")]
bad_total = 0
for mission, dtype, B, pattern in SHAPES:
    air = BN.AIRCRAFT5 if mission == "mixed" else ("tempest",)
    bt = tol_amd.Batch(mission, air, ts=200, dtype=dtype, pattern=pattern)
    bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, mission, len(air)))
    dXs, dF, dG = BN.make_inputs(bt, torch, B, 0, 4)
    obj = torch.zeros(B, dtype=dF.dtype, device="cuda")
    it = torch.int64 if dtype == "f64" else torch.int32

    def checksum():
        return torch.stack([dF.view(it).sum(dtype=torch.int64), dG.view(it).sum(dtype=torch.int64), obj.view(it).sum(dtype=torch.int64)])

    ref = []
    for x in dXs:
        dF.zero_(); dG.zero_(); obj.zero_()
        bt.eval(x, dF, dG, obj=obj, B=B)
        ref.append(checksum())
    ref = torch.stack(ref)
    n, bad, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < SECONDS:
        sums = []
        for i in range(200):
            bt.eval(dXs[i % 4], dF, dG, obj=obj, B=B)
            sums.append(checksum())
        got = torch.stack(sums)
        want = ref[torch.arange(200, device="cuda") % 4]
        bad += int((got != want).any(dim=1).sum().item())
        n += 200
    print(f"{mission:5s} {dtype} B={B:5d} {pattern:9s}: {n} evaluations in {time.perf_counter() - t0:.1f} s, {bad} differ from the first evaluation of their X buffer", flush=True)
    bad_total += bad
    bt.close()

# Streams.  (a) one batch object evaluated on two streams in turn, same F / G: the library owns per-launch workspace and orders a
# launch behind the previous one when the stream changes (include/tolfg.h, "Stream contract"); (b) two batch objects, each on
# its own stream with its own buffers, in flight together.
for label in ("one batch, two streams in turn", "two batches, two streams, in flight together"):
    B = 2048
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sets = []
    for k in range(1 if label.startswith("one") else 2):
        bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=200, dtype="f64")
        bt.set_trajectories(BN.make_trajectories(tol_amd, B, 0, "mixed", 5))
        dXs, dF, dG = BN.make_inputs(bt, torch, B, 0, 4)
        obj = torch.zeros(B, dtype=dF.dtype, device="cuda")
        ref = []
        for x in dXs:
            bt.eval(x, dF, dG, obj=obj, B=B)
            ref.append(torch.stack([dF.view(torch.int64).sum(), dG.view(torch.int64).sum(), obj.view(torch.int64).sum()]))
        sets.append((bt, dXs, dF, dG, obj, torch.stack(ref)))
    torch.cuda.synchronize()
    n, bad, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < SECONDS / 2:
        sums = []
        for i in range(200):
            st = s1 if i % 2 == 0 else s2
            bt, dXs, dF, dG, obj, ref = sets[(i % 2) % len(sets)]
            with torch.cuda.stream(st):
                bt.eval(dXs[(i // 2) % 4], dF, dG, obj=obj, B=B)
                sums.append((torch.stack([dF.view(torch.int64).sum(), dG.view(torch.int64).sum(), obj.view(torch.int64).sum()]), ref[(i // 2) % 4]))
        torch.cuda.synchronize()
        bad += sum(int((g != w).any().item()) for g, w in sums)
        n += 200
    print(f"{label}: {n} evaluations in {time.perf_counter() - t0:.1f} s, {bad} differ from the first evaluation of their X buffer", flush=True)
    bad_total += bad
    for t in sets:
        t[0].close()

# The SNOPT callback (host x -> host F, G), single trajectory: F and G are poisoned before every call, so an entry a call did
# not write (a finalizing wave lost, a completion word seen before the data) shows as well as a wrong one.
import numpy as np

for ts, staged in ((200, False), (200, True), (2000, False)):
    p = tol_amd.Problem("S10", "tempest", ts=ts, persistent_arrays=not staged)
    x0 = p.x0()
    rng = np.random.default_rng(5)
    xs = [x0 * (1.0 + 0.03 * rng.uniform(-1, 1, x0.shape)) for _ in range(4)]
    x, F, G = np.zeros(p.n), np.zeros(p.neF), np.zeros(p.neG)
    refs = []
    for xv in xs:
        f, g, st = p.define_fg(xv)
        assert st == 1
        refs.append((f.copy(), g.copy()))
    if not staged:
        p.register_arrays(x, F, G)
    n, bad, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < SECONDS / 2:
        for i in range(100):
            np.copyto(x, xs[i % 4])
            F.fill(np.nan); G.fill(np.nan)
            _, _, st = p.define_fg(x, F=F, G=G)
            if st != 1 or not (np.array_equal(F, refs[i % 4][0]) and np.array_equal(G, refs[i % 4][1])):
                bad += 1
        n += 100
    print(f"callback S10 ts={ts:4d} {'staged  ' if staged else 'in place'}: {n} calls in {time.perf_counter() - t0:.1f} s, {bad} differ from the first call with their x", flush=True)
    bad_total += bad
    if not staged:
        p.forget_arrays()
    p.close()
sys.exit(1 if bad_total else 0)
