#!/usr/bin/env python3
"""A/B of launch-plan variants on the SAME buffers in ONE process: each variant is a batch object created under its
environment overrides (TOLFG_* measurement switches are read at batch creation); the variants are timed in turn, several
rounds, so that box, clocks and buffer placement are common to all of them.  Outputs are compared bitwise with the first.

usage: plan_ab.py MISSION DTYPE BATCH [--ts N] [--pattern P] [--rounds R] name[:K=V[,K=V...]] ...
  e.g. plan_ab.py S10 f64 1024 base sub32:TOLFG_SUB_NODES=32 tail:TOLFG_TAIL=256:40"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("TOLFG_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tol_amd", "lib", "libtolfg_measure.so"))   # the TOLFG_* switches exist in the measurement build only (tol_amd/csrc/knobs.h)
import tol_amd
import bench as BN

ap = argparse.ArgumentParser()
ap.add_argument("mission")
ap.add_argument("dtype")
ap.add_argument("batch", type=int)
ap.add_argument("variants", nargs="+")
ap.add_argument("--ts", type=int, default=200)
ap.add_argument("--pattern", default="reference")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
B, ts = args.batch, args.ts
air = BN.AIRCRAFT5 if args.mission == "mixed" else ("tempest",)
trajs = BN.make_trajectories(tol_amd, B, 0, args.mission, len(air))
KNOWN = [k for k in os.environ if k.startswith("TOLFG_")]


def make(spec):
    name, _, envs = spec.partition(":")
    for k in list(os.environ):
        if k.startswith("TOLFG_") and k not in KNOWN:
            del os.environ[k]
    if envs:
        for kv in envs.split(","):
            k, _, v = kv.partition("=")
            os.environ[k] = v
    bt = tol_amd.Batch(args.mission, air, ts=ts, dtype=args.dtype, pattern=args.pattern)
    bt.set_trajectories(trajs)
    return name, bt


batches = [make(v) for v in args.variants]
dXs, dF, dG = BN.make_inputs(batches[0][1], torch, B, 0, 4)
alg = batches[0][1].algorithmic_bytes(B)
ref = None
for name, bt in batches:
    dF.zero_(); dG.zero_()
    bt.eval(dXs[0], dF, dG, B=B)
    torch.cuda.synchronize()
    if ref is None:
        ref = (dF.clone(), dG.clone())
    else:
        same = torch.equal(dF, ref[0]) and torch.equal(dG, ref[1])
        print(f"{name}: outputs {'bitwise equal to' if same else 'DIFFER from'} {batches[0][0]}", flush=True)
BN.settle(lambda i: batches[0][1].eval(dXs[i % 4], dF, dG, B=B), torch.cuda.synchronize, 5)
res = {n: [] for n, _ in batches}
for rnd in range(args.rounds):
    for name, bt in batches:
        for i in range(5):
            bt.eval(dXs[i % 4], dF, dG, B=B)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.reps):
            bt.eval(dXs[i % 4], dF, dG, B=B)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(1e3 * e0.elapsed_time(e1) / args.reps)
print(f"## {args.mission} {args.dtype} B={B} ts={ts} pattern={args.pattern}: us per evaluation, {args.rounds} rounds of {args.reps} launches, variants in turn on the same buffers")
for name, v in res.items():
    best = min(v)
    print(f"{name:28s} " + "  ".join(f"{t:7.2f}" for t in v) + f"   min {best:7.2f} us = {alg / best / 8e6:5.3f} of peak", flush=True)
