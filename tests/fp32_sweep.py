#!/usr/bin/env python3
"""BASELINE configs[4] / SURVEY.md section 8(d) config 5: mixed G7+S10 batch of 8192 trajectories
(mission = b mod 2, air-frame = b mod 5 over all five .param files, ts = 200), evaluated on the GPU
in fp64 and in fp32; reports max abs / max scaled error of fp32 against fp64 per row class, and of
fp64 against the CPU oracle on a subset.  Run on the GPU box; writes a markdown table to stdout.
(Kept under tests/ because it checks against oracle/, which only test code may use.)

scaled error = |a - b| / (1 + |b|), the measure the parity tests use.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

AIRCRAFT = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]


def row_classes_F(N, nb):
    """class name -> boolean mask over F"""
    neF = 8 * N + 1 + nb
    idx = np.arange(neF)
    r = (idx - 1) % 8 + 1
    dyn = (idx >= 1) & (idx < 8 * N + 1)
    return {"objective": idx == 0,
            "defect x,y,z (r1-r3)": dyn & (r <= 3),
            "defect Va,gam,chi (r4-r6)": dyn & (r >= 4) & (r <= 6),
            "defect phi,CL (r7-r8)": dyn & (r >= 7),
            "boundary rows": idx >= 8 * N + 1}


def row_classes_G(iG, N, nb):
    r = (iG - 1) % 8 + 1
    dyn = (iG >= 1) & (iG < 8 * N + 1)
    return {"objective row (gradient entries)": iG == 0,
            "defect x,y,z (r1-r3)": dyn & (r <= 3),
            "defect Va,gam,chi (r4-r6)": dyn & (r >= 4) & (r <= 6),
            "defect phi,CL (r7-r8)": dyn & (r >= 7),
            "boundary rows": iG >= 8 * N + 1}


def run(B, N, seed0):
    """One mixed batch (mission = b mod 2, air-frame = b mod 5), evaluated by ONE launch per element type."""
    import torch
    import tol_amd
    from oracle import oracle as O
    rng = np.random.default_rng(seed0)
    trajs = []
    for t in range(B):
        ms = ("S10", "G7")[t % 2]
        trajs.append(tol_amd.Trajectory(aircraft=t % 5, mission=ms, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20),
                                        radius_goal=100.0 if ms == "S10" else 0.0,
                                        xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50), zi=-40.0))
    b64 = tol_amd.Batch("mixed", AIRCRAFT, ts=N, dtype="f64")
    b32 = tol_amd.Batch("mixed", AIRCRAFT, ts=N, dtype="f32")
    b64.set_trajectories(trajs)
    b32.set_trajectories(trajs)
    X = np.zeros((B, b64.n))
    for t in range(B):
        r2 = np.random.default_rng(seed0 + 1 + t)
        x = b64.x0(t, zi=-40.0)
        x = x + 0.05 * r2.uniform(-1, 1, x.shape) * (1 + np.abs(x))
        nd = x[1:].reshape(N + 1, 11)
        nd[:, 3] = r2.uniform(12, 18, N + 1)
        nd[:, 10] = r2.uniform(5, 15, N + 1)
        x[0] = abs(x[0]) + 0.01
        X[t] = x
    X32 = X.astype(np.float32)
    res = {}
    for name, bt, Xh in (("f64", b64, X32.astype(np.float64)), ("f32", b32, X32)):
        dX, dF, dG = bt.alloc(B)
        dX[:, :bt.n] = torch.from_numpy(Xh).cuda()
        bt.eval(dX, dF, dG)
        torch.cuda.synchronize()
        res[name] = (dF.double().cpu().numpy(), dG.double().cpu().numpy())
    out = []
    for mission, off in (("S10", 0), ("G7", 1)):
        n, neF, neG = b64.sizes_of(mission)
        iG, _ = b64.pattern(mission)
        nb = 11 if mission == "S10" else 12
        F64, G64 = res["f64"][0][off::2, :neF], res["f64"][1][off::2, :neG]
        F32, G32 = res["f32"][0][off::2, :neF], res["f32"][1][off::2, :neG]
        rows = []
        for kind, A, Bm, classes in (("F", F32, F64, row_classes_F(N, nb)), ("G", G32, G64, row_classes_G(iG, N, nb))):
            for cname, m in classes.items():
                a, b = A[:, m], Bm[:, m]
                rows.append((mission, kind, cname, float(np.abs(a - b).max()), float((np.abs(a - b) / (1 + np.abs(b))).max()),
                             float(np.abs(b).max())))
        # fp64 against the oracle on a subset (same float32-rounded inputs)
        worst = 0.0
        idx = list(range(off, B, 2))
        for t in idx[::max(1, len(idx) // 64)]:
            tr = trajs[t]
            o = O.Problem(mission, AIRCRAFT[tr.aircraft], N=N, east_goal=tr.east_goal, north_goal=tr.north_goal,
                          radius_goal=tr.radius_goal, start=(tr.xi, tr.yi, -40.0), Vref=tr.Vref, href=tr.href)
            Fo, Go = o.eval(X32[t].astype(np.float64))
            m = o.undefined_mask()
            eF = (np.abs(res["f64"][0][t, :neF] - Fo) / (1 + np.abs(Fo))).max()
            eG = np.where(m, 0, np.abs(res["f64"][1][t, :neG] - Go) / (1 + np.abs(Go))).max()
            worst = max(worst, eF, eG)
        out.append((mission, rows, worst))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--ts", type=int, default=200)
    a = ap.parse_args()
    print(f"# fp32 vs fp64 sweep: ONE mixed batch of {a.batch} trajectories (S10: even b, G7: odd b), air-frame = b mod 5, ts = {a.ts}, one launch per element type\n")
    print("Inputs are rounded to float32 first, so both runs see identical x. scaled = |f32 - f64| / (1 + |f64|).\n")
    print("| mission | array | row class | max abs err | max scaled err | max abs value |\n|---|---|---|---|---|---|")
    for mission, rows, worst in run(a.batch, a.ts, 11):
        for r in rows:
            print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.3e} | {r[4]:.3e} | {r[5]:.3e} |")
        print(f"| {mission} | F,G | fp64 HIP vs CPU oracle, 64 trajectories | | {worst:.3e} | |")


if __name__ == "__main__":
    main()
