import os, sys
sys.path.insert(0, "/root/repo")
import os as _os
_os.environ.setdefault("TOLFG_LIBRARY", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tol_amd", "lib", "libtolfg_measure.so"))   # the TOLFG_* switches exist in the measurement build only (tol_amd/csrc/knobs.h)
import torch, tol_amd
import bench as BN
B, ts = 8192, 200
trajs = BN.make_trajectories(tol_amd, B, 0, "mixed", 5)
def batch(dtype, env):
    for k in ("TOLFG_WAVES_PER_CU", "TOLFG_TILE_NODES", "TOLFG_NT_STORES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bt = tol_amd.Batch("mixed", BN.AIRCRAFT5, ts=ts, dtype=dtype)
    bt.set_trajectories(trajs)
    return bt
b64 = batch("f64", {})
dXs64, dF64, dG64 = BN.make_inputs(b64, torch, B, 0, 4)
b32 = batch("f32", {})
dXs32, dF32, dG32 = BN.make_inputs(b32, torch, B, 0, 4)
# the fp32 G as a view into the fp64 G buffer as well: the same physical memory
G32in64 = dG64.view(torch.float32)[:, : dG32.shape[1]]
G32in64 = torch.as_strided(dG64.view(torch.float32).view(-1), (B, dG32.shape[1]), (dG32.shape[1], 1))
def ev(bt, dXs, dF, G, reps=30):
    for i in range(5): bt.eval(dXs[i % 4], dF, G, B=B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): bt.eval(dXs[i % 4], dF, G, B=B)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
g, st = BN.store_shape_rate(b64, torch, dXs64, dF64, dG64, B, ts, 104, reps=20)
print(f"fp64 default plan: store loop {st:.1f} us = {g:.0f} GB/s; evaluation {ev(b64, dXs64, dF64, dG64):.1f} us", flush=True)
for env in ({}, {"TOLFG_WAVES_PER_CU": "8"}, {"TOLFG_WAVES_PER_CU": "16"}, {"TOLFG_WAVES_PER_CU": "0"}, {"TOLFG_TILE_NODES": "64"}, {"TOLFG_TILE_NODES": "64", "TOLFG_WAVES_PER_CU": "8"},
            {"TOLFG_TILE_NODES": "64", "TOLFG_WAVES_PER_CU": "16"}, {"TOLFG_TILE_NODES": "96"}, {"TOLFG_TILE_NODES": "72"}):
    bt = batch("f32", env)
    for name, G in (("own buffer", dG32), ("inside the fp64 buffer", G32in64)):
        g, st = BN.store_shape_rate(bt, torch, dXs32, dF32, G, B, ts, 104, reps=20)
        print(f"fp32 {str(env):62s} {name:24s}: store loop {st:6.1f} us = {g:5.0f} GB/s; evaluation {ev(bt, dXs32, dF32, G):6.1f} us", flush=True)
    bt.close()
