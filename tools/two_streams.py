#!/usr/bin/env python3
"""Side measurement: independent batches of the headline shape in flight on several HIP streams (bench.two_streams_record
for 2 x B, and splits of ONE 4096-trajectory evaluation into concurrent pieces)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, tol_amd

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=100); a = ap.parse_args()
args = argparse.Namespace(mission="S10", aircraft="tempest", ts=200, dtype="f64")
for B in (4096, 2048, 1024):
    r = bench.two_streams_record(tol_amd, torch, args, B, 0, a.steps)
    print("2 streams x B=%d: %.1f us per evaluation of B, i.e. %.1f us per 2B; %.3f of peak" % (B, r["us_per_evaluation"], 2 * r["us_per_evaluation"], r["frac_of_hbm_peak"]))


def pieces(sizes, steps):
    """ONE evaluation of sum(sizes) trajectories issued as len(sizes) concurrent launches; wall time per evaluation with a
    join (all streams synchronised) after every evaluation, as a caller of one evaluation would see it."""
    streams = [torch.cuda.Stream() for _ in sizes]
    sets, first = [], 0
    for B in sizes:
        bt = tol_amd.Batch("S10", ("tempest",), ts=200)
        bt.set_trajectories(bench.make_trajectories(tol_amd, B, first, "S10", 1)); first += B
        dXs, dF, dG = bench.make_inputs(bt, torch, B, first, 2)
        sets.append((bt, dXs, dF, dG))
    ev = [torch.cuda.Event() for _ in sizes]
    def one(i):
        for k, (bt, dXs, dF, dG) in enumerate(sets):
            with torch.cuda.stream(streams[k]):
                bt.eval(dXs[i % 2], dF, dG)
                ev[k].record()
        for k in range(1, len(sizes)):           # join on stream 0, then every stream waits for the join
            streams[0].wait_event(ev[k])
        j = torch.cuda.Event(); 
        with torch.cuda.stream(streams[0]): j.record()
        for k in range(1, len(sizes)): streams[k].wait_event(j)
    for i in range(5): one(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): one(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("one evaluation of %d as pieces %s: %.1f us" % (sum(sizes), sizes, 1e6 * dt / steps))
    for s in sets: s[0].close()

pieces([4096], a.steps)
pieces([2048, 2048], a.steps)
pieces([2560, 1536], a.steps)
pieces([3072, 1024], a.steps)
pieces([1366, 1365, 1365], a.steps)
pieces([1024, 1024, 1024, 1024], a.steps)
