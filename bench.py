#!/usr/bin/env python3
"""bench.py -- collocation-node F/G+Jacobian evaluations per second on 1..8 MI355X.

A step is ONE pass of the hot path over one device-resident batch: a single fused launch that
evaluates F and G (objective, defects, boundary rows and the whole sparse Jacobian) of B
trajectories per GPU, plus the gather of the per-trajectory objectives (on N > 1 GPUs that gather
is one RCCL all-gather over xGMI, issued asynchronously so it overlaps the next step's launch).
Inputs are resident in HBM before the timed region starts.

Workload (BASELINE.json configs[1] batched as configs[3] prescribes): problemS10, tempest.param,
ts = 200 collocation nodes, fp64; per-trajectory linear-shear wind Vref~U(0,5), href~U(5,20),
start offset (xi,yi,zi)~U(-50,50)^2 x U(-100,-20), x = x0(start) + 5 % noise, seed 1000 + global
trajectory index.  Weak scaling: B per GPU is fixed.

Launch: `python bench.py --gpus 1 ...` or, for N > 1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W`.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def make_inputs(bt, tol_amd, B, first_index):
    """Synthetic batch: trajectory table + X rows (host, float64)."""
    trajs, X = [], np.empty((B, bt.n))
    zis = np.empty(B)
    for t in range(B):
        rng = np.random.default_rng(1000 + first_index + t)
        tr = tol_amd.Trajectory(aircraft=0, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20),
                                north_goal=0.0, east_goal=400.0, radius_goal=100.0,
                                xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50))
        zis[t] = rng.uniform(-100, -20)
        trajs.append(tr)
    bt.set_trajectories(trajs)
    for t in range(B):
        rng = np.random.default_rng(5000000 + first_index + t)
        x = bt.x0(t, zi=zis[t])
        X[t] = x + 0.05 * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))
        X[t, 0] = abs(X[t, 0]) + 0.01
    return trajs, X


def cpu_baseline(args, seconds):
    """Time the oracle (kind 'port') on this box's host cores; rank 0, N = 1 only.
    Main figure: ONE core, the reference's own evaluation order (one Jacobian entry per call, its
    whole row rebuilt each time: src/problem.cpp:782-806,1035-1208), -O2.  Extras: the same at -O0
    (the reference's shipped build config, .cproject:30-31) and the fused per-node form on all cores."""
    from oracle import oracle as O
    o = O.Problem(args.mission, args.aircraft, N=args.ts)
    x = O.perturbed(o, 7)
    disp = o.dispatch()
    out = {}
    for opt in ("O2", "O0"):
        o.eval_entrywise(x, opt=opt, dispatch=disp)
        n, t0 = 0, time.perf_counter()
        budget = seconds if opt == "O2" else seconds / 3
        while time.perf_counter() - t0 < budget:
            o.eval_entrywise(x, opt=opt, dispatch=disp)
            n += 1
        dt = time.perf_counter() - t0
        out[opt] = (n * args.ts / dt, n, dt)
    # fused per-node form on one core (variant b)
    o.eval(x)
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds / 6:
        o.eval(x)
        n1 += 1
    fused1 = n1 * args.ts / (time.perf_counter() - t0)
    # fused per-node form, all host cores, OpenMP over trajectories
    # worker pool sized to the GPU box's CPU share (16 per GPU), not to every core the host shows
    cores = int(os.environ.get("TOLFG_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    Bc = max(cores * 8, 64)
    probs = [o] * Bc
    X = np.stack([O.perturbed(o, 100 + i) for i in range(Bc)])
    O.eval_batch(probs, X, nthreads=cores)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds / 2:
        _, _, used = O.eval_batch(probs, X, nthreads=cores)
        n += 1
    dt = time.perf_counter() - t0
    fused = n * Bc * args.ts / dt
    base = {"value": out["O2"][0], "unit": "node-evals/s", "cores": 1, "kind": "port",
            "sample": f"{out['O2'][1]} evaluations of one {args.mission}/{args.aircraft}/ts={args.ts} trajectory "
                      f"in {out['O2'][2]:.1f} s, reference evaluation order (entry-wise Jacobian), gcc -O2",
            "value_O0": out["O0"][0],
            "fused_one_core": {"value": fused1, "cores": 1, "sample": f"{n1} evaluations, per-node fused form, gcc -O2"},
            "fused_all_cores": {"value": fused, "cores": used, "sample": f"{n} x {Bc} trajectories, per-node fused form, OpenMP"}}
    return base


def callback_mode(tol_amd, mission, aircraft, ts, calls):
    """Single-trajectory SNOPT-callback rate: host x -> host F, G through DEFINEGusrfg_, entered the
    way snOptA does (all arguments by reference, prepared once: the loop times the C ABI, not the
    construction of ctypes objects)."""
    import ctypes as C
    p = tol_amd.Problem(mission, aircraft, ts=ts)
    p.make_current()
    L = tol_amd.lib()
    dp = C.POINTER(C.c_double)
    x = np.ascontiguousarray(p.x0())
    F, G = np.zeros(p.neF), np.zeros(p.neG)
    st, n, neF, neG = C.c_int(1), C.c_int(p.n), C.c_int(p.neF), C.c_int(p.neG)
    one, zero = C.c_int(1), C.c_int(0)
    argv = (C.byref(st), C.byref(n), x.ctypes.data_as(dp), C.byref(one), C.byref(neF), F.ctypes.data_as(dp),
            C.byref(one), C.byref(neG), G.ctypes.data_as(dp), None, C.byref(zero), None, C.byref(zero), None, C.byref(zero))
    fn = L.DEFINEGusrfg_
    for _ in range(50):
        fn(*argv)
    t0 = time.perf_counter()
    for _ in range(calls):
        fn(*argv)
    dt = time.perf_counter() - t0
    assert st.value == 1 and np.isfinite(F).all()
    p.close()
    return {"workload": f"{mission}/{aircraft}/ts={ts} single trajectory, DEFINEGusrfg_ host->host",
            "us_per_call": 1e6 * dt / calls, "node_evals_per_s": calls * ts / dt, "calls": calls}


def compact_side_run(tol_amd, torch, args, dX, B, steps=50):
    """SURVEY.md section 8(f) rank 1, measured beside the headline: the same batch through the compact
    sparsity pattern (46 instead of 104 entries per node).  Not the headline metric -- it changes
    the pattern handed to SNOPT."""
    bc = tol_amd.Batch(args.mission, (args.aircraft,), ts=args.ts, dtype=args.dtype, device=dX.device.index,
                       pattern="compact")
    # same trajectories as the headline batch
    import ctypes as C
    from tol_amd import capi
    make_inputs(bc, tol_amd, B, first_index=0)
    _, dF, dG = bc.alloc(B)
    obj = torch.empty(B, dtype=dF.dtype, device=dF.device)
    for _ in range(5):
        bc.eval(dX, dF, dG, obj=obj)
    torch.cuda.synchronize()
    bc.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        bc.eval(dX, dF, dG, obj=obj)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _, kms, _ = bc.kernel_time()
    alg = bc.algorithmic_bytes(B)
    return {"value": B * args.ts * steps / dt, "unit": "node-evals/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "kernel_ms": kms, "algorithmic_bytes_per_launch": alg, "achieved_GBs": alg / (kms * 1e-3) / 1e9,
            "frac_of_hbm_peak": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_node": alg / (B * args.ts)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    ap.add_argument("--ts", type=int, default=200)
    ap.add_argument("--mission", default="S10")
    ap.add_argument("--aircraft", default="tempest")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-callback", action="store_true")
    ap.add_argument("--x-buffers", type=int, default=4,
                    help="distinct input batches rotated over the steps; 4 x 72 MB exceed the 256 MiB Infinity "
                         "Cache, so every step reads its X from HBM rather than from a cache that kept it")
    ap.add_argument("--ld-pad", type=int, default=0, help="pad the row strides of X, F, G to this many elements (0 = 16 bytes)")
    ap.add_argument("--pattern", default="reference", choices=["reference", "compact"],
                    help="Jacobian sparsity pattern; the headline metric is quoted on the reference's own pattern")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1; gloo (objectives staged through host memory) only "
                         "rehearses the multi-rank step loop when several ranks must share one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tol_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local = local % max(ndev, 1)               # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            try:      # eager communicator creation on this rank's GPU (RCCL over xGMI)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            except TypeError:
                dist.init_process_group("nccl")
            # first collective here, outside any timing: RCCL builds its rings lazily; if it cannot
            # (driver / IPC trouble) the run falls back to gloo rather than losing the scaling point
            try:
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as exc:      # noqa: BLE001
                sys.stderr.write(f"bench.py: RCCL collective failed ({exc}); falling back to gloo\n")
                dist.destroy_process_group()
                dist.init_process_group("gloo")
                args.backend = "gloo"
        else:
            dist.init_process_group("gloo")

    B = args.batch
    bt = tol_amd.Batch(args.mission, (args.aircraft,), ts=args.ts, dtype=args.dtype, device=local, pattern=args.pattern)
    _, X = make_inputs(bt, tol_amd, B, first_index=rank * B)
    dX, dF, dG = bt.alloc(B, pad=(args.ld_pad or None))
    dX[:, :bt.n] = torch.from_numpy(X).to(bt.torch_dtype()).cuda()
    del X
    # further input batches: the same trajectories, the decision vectors rotated by whole rows
    # (every row is a valid input of ITS trajectory's shape; values differ from step to step)
    dXs = [dX] + [torch.roll(dX, shifts=7 * (j + 1), dims=0).contiguous() for j in range(max(args.x_buffers, 1) - 1)]
    obj = [torch.empty(B, dtype=dF.dtype, device=dF.device) for _ in range(2)]
    gdev = dF.device if args.backend == "nccl" else torch.device("cpu")
    allobj = [torch.empty(B * world, dtype=dF.dtype, device=gdev) for _ in range(2)] if world > 1 else None
    pending = [None, None]

    def step(i):
        s = i & 1
        if world > 1 and pending[s] is not None:
            pending[s].wait()                      # buffer reuse: the gather of step i-2 must be done
        bt.eval(dXs[i % len(dXs)], dF, dG, obj=obj[s])   # finalize_kernel also writes the objectives, contiguous
        if world > 1:
            src = obj[s] if args.backend == "nccl" else obj[s].cpu()
            pending[s] = dist.all_gather_into_tensor(allobj[s], src, async_op=True)

    def fence():
        for s in (0, 1):
            if pending[s] is not None:
                pending[s].wait()
                pending[s] = None
        torch.cuda.synchronize()
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events on the launch stream around the dominant kernel of every timed step (recorded
    # inside the library, include/tolfg.h: tolfg_batch_set_timing)
    bt.set_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    nlaunch, kern_ms, kern_min_ms = bt.kernel_time()
    bt.set_timing(False)
    assert nlaunch == args.steps
    el = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = el[0].item(), el[1].item()

    # sanity: the objectives that came back are finite and, on N > 1, every rank's shard arrived
    final = allobj[(args.steps - 1) & 1] if world > 1 else obj[(args.steps - 1) & 1]
    assert torch.isfinite(final).all(), "non-finite objective in the gathered result"
    if world > 1:   # this rank's shard sits at its place in the gathered vector
        mine = obj[(args.steps - 1) & 1].to(final.device)
        assert torch.equal(final[rank * B:(rank + 1) * B], mine), "gathered objectives are out of order"

    if rank == 0:
        nodes_per_step = B * world * args.ts
        value = nodes_per_step * args.steps / elapsed
        alg_bytes = bt.algorithmic_bytes(B)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    tj = json.load(fh)
                if tj.get("batch") == B and tj.get("ts") == args.ts and tj.get("dtype") == args.dtype:
                    traffic = tj.get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        line = {
            "metric": "collocation-node F/G+Jacobian evals/sec",
            "value": value, "unit": "node-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"problem{args.mission}/{args.aircraft}.param/ts={args.ts} (BASELINE configs[1]) as a "
                                   f"device-resident batch of {B} trajectories per GPU with randomized shear wind and "
                                   f"start offsets (configs[3] recipe); one fused F+G launch + objective gather per step",
                       "mission": args.mission, "aircraft": args.aircraft, "ts": args.ts, "pattern": args.pattern,
                       "batch_per_gpu": B, "global_batch": B * world, "x_buffers": len(dXs),
                       "parallelism": (f"batch-sharded x{world}, " + ("RCCL" if args.backend == "nccl" else "gloo (host-staged)") +
                                       " all-gather of objectives") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "tolfg::fg_kernel", "kernel_ms": kern_ms, "kernel_min_ms": kern_min_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_node": alg_bytes / (B * args.ts)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        if world == 1 and args.pattern == "reference" and not args.no_callback:
            line["next_compact_pattern"] = compact_side_run(tol_amd, torch, args, dX, B)
        if world == 1 and not args.no_callback:
            line["callback"] = [callback_mode(tol_amd, "S10", "tempest", 200, 300),
                                callback_mode(tol_amd, "S10", "skywalker", 2000, 100)]
        print(json.dumps(line), flush=True)

    if world > 1:
        fence()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
