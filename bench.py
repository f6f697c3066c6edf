#!/usr/bin/env python3
"""bench.py -- collocation-node F/G+Jacobian evaluations per second on 1..8 MI355X.

A step is ONE pass of the hot path over one device-resident batch: a single launch that evaluates F
and G (objective, defects, boundary rows and the whole sparse Jacobian) of B trajectories per GPU,
plus the gather of the per-trajectory objectives (on N > 1 GPUs that gather is one RCCL all-gather
over xGMI, issued asynchronously so it overlaps the next step's launch).  Inputs are resident in HBM
before the timed region starts.

Headline workload = the largest BASELINE config that fits one GPU, configs[4]: a mixed batch of
problemG7 + problemS10 trajectories (mission = b mod 2), all five aircraft .param files (b mod 5),
ts = 200 collocation nodes, fp64, 8192 trajectories per GPU; per-trajectory linear-shear wind
Vref~U(0,5), href~U(5,20), start offset (xi,yi,zi)~U(-50,50)^2 x U(-100,-20), x = x0(start) + 5 %
noise, seed 1000 + global trajectory index.  Weak scaling by
default (--batch trajectories per GPU); --global-batch G fixes the total instead (strong scaling:
rank r owns shard_bounds(G, r, N)).

The JSON line also carries `configs`: one measured record per BASELINE config.  On any number of
GPUs: configs[3] (global batch 1024, S10) and configs[4] (global batch 8192 mixed, fp64 and fp32) AS
STATED, i.e. strong scaling -- every rank evaluates its shard and the objectives are all-gathered,
inside the same process group after the headline.  On one GPU also the single-trajectory configs
(device-resident and through the host->host SNOPT callback), the shares one of 8 GPUs gets, and the
side records (the round-2 headline shape S10/4096, compact pattern, two streams).

Launch: `python bench.py --gpus N ...` as a plain command (for N > 1 it starts its own N ranks as a child
`python -m torch.distributed.run`, before anything touches a GPU, and relays rank 0's line and the exit code) or,
for N > 1, directly as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N --steps K --warmup W`.  Rank 0 prints ONE JSON line.

Every timed region is preceded by at least --warmup steps AND at least 0.25 s of back-to-back launching (a fresh box
ramps its clocks over the first milliseconds; `warmup_steps_run` says how many steps that took), and followed by a
calibration of the box in the same process: the vendor's fill kernel and the bare store loop of the launch's own
shape (`roofline.box_fill_GBs`, `box_stream_shape_GBs`, `frac_of_box_fill`, `vs_bare_store_loop`), so that a
slow box can be told from a slow kernel.  Both are reference points measured beside the evaluation, not ceilings: the
evaluation overlaps its loads with its stores and has read 1.00-1.03 x the bare store loop of its own shape.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
AIRCRAFT5 = ("tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will")
MIN_WARM_S = 0.25         # every timed region is preceded by at least this much back-to-back launching (and >= --warmup steps):
                          # a fresh box is still ramping its clocks during the first milliseconds (profiles/r03_clocks_power.md)
CALIBRATE = True          # --no-calibration: the calibration launches are skipped (NaN in their fields)
_BOX = {}                 # per-process calibration of this box (box_fill)


def settle(step, fence, warmup, max_over_ranks=None):
    """Warm-up before a timed region: steps are run back to back, in blocks, until at least `warmup` of them AND at least
    MIN_WARM_S seconds of launching have gone by.  The block sizes are agreed over the ranks (every rank runs the same
    number of steps: a step may hold a collective).  Returns the number of steps run."""
    n, t_start, block = 0, time.perf_counter(), max(int(warmup), 3)
    while True:
        t0 = time.perf_counter()
        for _ in range(block):
            step(n)
            n += 1
        fence()
        now = time.perf_counter()
        remaining = MIN_WARM_S - (now - t_start)
        nxt = 0 if remaining <= 0 else max(1, int(math.ceil(remaining / max((now - t0) / block, 1e-7))))
        if max_over_ranks is not None:
            nxt = int(max_over_ranks([float(nxt)])[0])
        if nxt == 0:
            return n
        block = min(nxt, 100000)


def box_fill(torch, device):
    """Calibration, once per process: the vendor's fill kernel (torch.Tensor.fill_, a hipMemset-class kernel) over 800 MB of
    this GPU's HBM -- what THIS box's write path gives the simplest possible stream (tools/fill_reference.py)."""
    if not CALIBRATE:
        return float("nan")
    if "fill_GBs" not in _BOX:
        n = 100 * 1000 * 1000
        a = torch.empty(n, dtype=torch.float64, device=device)
        for _ in range(5):
            a.fill_(1.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            a.fill_(2.5)
        e1.record()
        torch.cuda.synchronize()
        _BOX["fill_GBs"] = 8.0 * n * 30 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a
        torch.cuda.empty_cache()
    return _BOX["fill_GBs"]


def store_shape_rate(bt, torch, dXs, dF, dG, B, ts, slab, reps=20):
    """Calibration: the bare store loop of THIS launch's shape (tolfg_batch_set_store_shape: the evaluation's grid, tile
    order, resident-wave cap and store flavour with only the Jacobian-slab stores in it), same buffers, same process.
    Returns (GB/s over the bytes it stores, us per launch).  G holds garbage afterwards (the next evaluation rewrites it)."""
    if not CALIBRATE:
        return float("nan"), float("nan")
    bt.set_store_shape(True)
    try:
        for _ in range(5):
            bt.eval(dXs[0], dF, dG, B=B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            bt.eval(dXs[0], dF, dG, B=B)
        e1.record()
        torch.cuda.synchronize()
    finally:
        bt.set_store_shape(False)
    us = 1e3 * e0.elapsed_time(e1) / reps
    stored = float(dG.element_size()) * B * ts * slab
    return stored / (us * 1e-6) / 1e9, us


def make_trajectories(tol_amd, B, first_index, mission="S10", n_aircraft=1):
    """The trajectory table of a synthetic batch (config 4 recipe; mission "mixed": b mod 2, air-frame b mod 5)."""
    trajs = []
    for t in range(B):
        g = first_index + t
        rng = np.random.default_rng(1000 + g)
        ms = ("S10", "G7")[g % 2] if mission == "mixed" else mission
        trajs.append(tol_amd.Trajectory(aircraft=g % n_aircraft, mission=ms, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20),
                                        north_goal=0.0, east_goal=400.0, radius_goal=100.0 if ms == "S10" else 0.0,
                                        xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50), zi=rng.uniform(-100, -20)))
    return trajs


def make_inputs(bt, torch, B, seed, buffers):
    """F and G: Batch.alloc -- G from the library's placement-probing allocator (tolfg_batch_alloc_outputs; the record's
    `placement` holds the candidates' probe times).  X buffers in HBM: initial guesses generated on the device (tolfg_batch_x0_device) + 5 % noise; the
    further buffers hold the same trajectories' vectors rotated by whole rows of the SAME mission/air-frame
    class (10 rows), so that every row stays a valid input of its trajectory and no step re-reads a cached X."""
    dX, dF, dG = bt.alloc(B)
    bt.x0_device(dX)
    gen = torch.Generator(device=dX.device).manual_seed(5000000 + seed)
    noise = torch.rand(dX.shape, dtype=dX.dtype, device=dX.device, generator=gen) * 2 - 1
    dt = dX[:, 0].clone()
    dX += 0.05 * noise * (1 + dX.abs())
    dX[:, 0] = dt.abs() + 0.01
    dXs = [dX] + [torch.roll(dX, shifts=10 * (j + 1), dims=0).contiguous() for j in range(max(buffers, 1) - 1)]
    for x in dXs[1:]:
        x[:, 0] = dX[:, 0]
    return dXs, dF, dG


def timed_evals(bt, torch, dXs, dF, dG, B, steps, warmup, obj=None):
    """`steps` evaluations back to back, twice.  Pass 1, uninstrumented: wall time, and the time between two HIP events
    recorded on the launch stream before the first and after the last launch, per launch (it contains the ~2 us between
    dependent launches, so it bounds the kernel's own duration from above).  Pass 2, instrumented: start / stop events
    attached to every dispatch (what rocprofv3 --kernel-trace also turns on) -- that costs 10-16 us per launch, part of
    it inside the reported duration (profiles/r02_event_cost.md), so it is reported beside, not as, the figure.
    Returns wall_s, per_launch_ms, instrumented_avg_ms, instrumented_min_ms."""
    settle(lambda i: bt.eval(dXs[i % len(dXs)], dF, dG, obj=obj, B=B), torch.cuda.synchronize, warmup)
    bt.set_timing(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        bt.eval(dXs[i % len(dXs)], dF, dG, obj=obj, B=B)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per_launch_ms = e0.elapsed_time(e1) / steps
    bt.set_timing(True)
    for i in range(steps):
        bt.eval(dXs[i % len(dXs)], dF, dG, obj=obj, B=B)
    torch.cuda.synchronize()
    n, avg_ms, min_ms = bt.kernel_time()
    bt.set_timing(False)
    assert n == steps
    return wall, per_launch_ms, avg_ms, min_ms


def device_record(tol_amd, torch, cfg, workload, mission, aircraft, ts, B, dtype, steps, device, x_buffers=4):
    """One BASELINE config, device-resident on one GPU: whole-step rate and HIP-event time per evaluation."""
    bt = tol_amd.Batch(mission, aircraft, ts=ts, dtype=dtype, device=device)
    bt.set_trajectories(make_trajectories(tol_amd, B, 0, mission, len(aircraft)))
    dXs, dF, dG = make_inputs(bt, torch, B, cfg, x_buffers)
    obj = torch.empty(B, dtype=dF.dtype, device=dF.device)
    wall, avg_ms, inst_ms, inst_min_ms = timed_evals(bt, torch, dXs, dF, dG, B, steps, 5, obj)
    assert torch.isfinite(obj).all()
    alg = bt.algorithmic_bytes(B)
    gbs = alg / (avg_ms * 1e-3) / 1e9
    rec = {"config": cfg, "workload": workload, "mode": "device-resident", "batch": B, "ts": ts, "dtype": dtype, "steps": steps,
           "ms_per_step": 1e3 * wall / steps, "node_evals_per_s": B * ts * steps / wall,
           "eval_us": 1e3 * avg_ms, "eval_us_instrumented": 1e3 * inst_ms, "eval_min_us_instrumented": 1e3 * inst_min_ms, "launches_per_step": 1,
           "algorithmic_bytes": alg, "achieved_GBs": gbs,
           "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "frac_of_box_fill": gbs / box_fill(torch, dF.device),
           "placement": getattr(bt, "placement", None)}
    if B > 8:       # callback-sized launches take the one-workgroup-per-trajectory kernel: no stream shape to calibrate
        shape_gbs, shape_us = store_shape_rate(bt, torch, dXs, dF, dG, B, ts, 104)
        rec.update(box_stream_shape_GBs=shape_gbs, box_stream_shape_us=shape_us, vs_bare_store_loop=gbs / shape_gbs)
    bt.close()
    del dXs, dF, dG
    torch.cuda.empty_cache()
    return rec


def cpu_baseline(args, seconds):
    """Time the oracle (kind 'port') on this box's host cores; rank 0, N = 1 only, on a bounded sample of the headline
    workload: one trajectory per (mission, air-frame) class of the batch, evaluated round-robin.
    Main figure: ONE core, the reference's own evaluation order (one Jacobian entry per call, its
    whole row rebuilt each time: src/problem.cpp:782-806,1035-1208), -O2.  Extras: the same at -O0
    (the reference's shipped build config, .cproject:30-31) and the fused per-node form on all cores."""
    from oracle import oracle as O
    if args.mission == "mixed":
        classes = [(("S10", "G7")[g % 2], AIRCRAFT5[g % 5]) for g in range(10)]
    else:
        classes = [(args.mission, args.aircraft)]
    probs = [O.Problem(m, a, N=args.ts, radius_goal=100.0 if m == "S10" else 0.0) for m, a in classes]
    xs = [O.perturbed(o, 7 + i) for i, o in enumerate(probs)]
    disp = [o.dispatch() for o in probs]
    out = {}
    for opt in ("O2", "O0"):
        probs[0].eval_entrywise(xs[0], opt=opt, dispatch=disp[0])
        n, t0 = 0, time.perf_counter()
        budget = seconds if opt == "O2" else seconds / 3
        while time.perf_counter() - t0 < budget:
            k = n % len(probs)
            probs[k].eval_entrywise(xs[k], opt=opt, dispatch=disp[k])
            n += 1
        dt = time.perf_counter() - t0
        out[opt] = (n * args.ts / dt, n, dt)
    # fused per-node form on one core (variant b)
    probs[0].eval(xs[0])
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds / 6:
        k = n1 % len(probs)
        probs[k].eval(xs[k])
        n1 += 1
    fused1 = n1 * args.ts / (time.perf_counter() - t0)
    # fused per-node form, all host cores, OpenMP over trajectories
    # worker pool sized to the GPU box's CPU share (16 per GPU), not to every core the host shows
    cores = int(os.environ.get("TOLFG_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
    Bc = max(cores * 8, 64) // len(probs) * len(probs)
    many = [probs[i % len(probs)] for i in range(Bc)]
    X = np.stack([O.perturbed(many[i], 100 + i) for i in range(Bc)])
    O.eval_batch(many, X, nthreads=cores)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds / 2:
        _, _, used = O.eval_batch(many, X, nthreads=cores)
        n += 1
    dt = time.perf_counter() - t0
    fused = n * Bc * args.ts / dt
    what = "+".join(sorted({m for m, _ in classes})) + "/" + ("five air-frames" if len(classes) > 1 else args.aircraft)
    base = {"value": out["O2"][0], "unit": "node-evals/s", "cores": 1, "kind": "port",
            "sample": f"{out['O2'][1]} evaluations over {len(probs)} {what}/ts={args.ts} trajectories (one per mission x air-frame class of "
                      f"the batch, round-robin) in {out['O2'][2]:.1f} s, reference evaluation order (entry-wise Jacobian), gcc -O2",
            "value_O0": out["O0"][0],
            "fused_one_core": {"value": fused1, "cores": 1, "sample": f"{n1} evaluations, per-node fused form, gcc -O2"},
            "fused_all_cores": {"value": fused, "cores": used, "sample": f"{n} x {Bc} trajectories, per-node fused form, OpenMP"}}
    return base


def callback_mode(tol_amd, mission, aircraft, ts, calls, cfg=None):
    """Single-trajectory SNOPT-callback rate: host x -> host F, G through DEFINEGusrfg_.  Main figure: the call
    entered from native code through an snFunA pointer, as snOptA enters it (tolfg_time_callback); beside it the
    same call entered from Python through ctypes (adds the foreign-call overhead of the test harness).  F and G
    are the same arrays every call, as SNOPT's are."""
    import ctypes as C
    p = tol_amd.Problem(mission, aircraft, ts=ts)
    x = np.ascontiguousarray(p.x0())
    us_native, F, G = p.time_callback(x, calls)
    assert np.isfinite(F).all() and np.isfinite(G).all() and F[0] != 0.0
    us_f_only, F2, _ = p.time_callback(x, calls, needG=False)      # what snOptA's line search asks for
    assert np.array_equal(F2, F)
    us_staged, F3, G3 = p.time_callback(x, calls, in_place=False)  # the default contract: every call staged
    assert np.array_equal(F3, F) and np.array_equal(G3, G)
    p.make_current()
    L = tol_amd.lib()
    dp = C.POINTER(C.c_double)
    st, n, neF, neG = C.c_int(1), C.c_int(p.n), C.c_int(p.neF), C.c_int(p.neG)
    one, zero = C.c_int(1), C.c_int(0)
    argv = (C.byref(st), C.byref(n), x.ctypes.data_as(dp), C.byref(one), C.byref(neF), F.ctypes.data_as(dp),
            C.byref(one), C.byref(neG), G.ctypes.data_as(dp), None, C.byref(zero), None, C.byref(zero), None, C.byref(zero))
    fn = L.DEFINEGusrfg_
    for _ in range(20):
        fn(*argv)
    t0 = time.perf_counter()
    for _ in range(calls):
        fn(*argv)
    dt = time.perf_counter() - t0
    assert st.value == 1
    p.close()
    rec = {"workload": f"{mission}/{aircraft}/ts={ts} single trajectory, DEFINEGusrfg_ host->host (PCIe inclusive)",
           "mode": "callback", "us_per_call": us_native, "node_evals_per_s": 1e6 * ts / us_native, "calls": calls,
           "entered_from": "native code through an snFunA pointer (tolfg_time_callback)",
           "arrays": "registered in place (tolfg_register_arrays / tolfg_config.persistent_arrays: the kernel reads x and writes "
                     "F, G in the caller's memory)",
           "us_per_call_needF_only": us_f_only,
           "us_per_call_staged": us_staged, "node_evals_per_s_staged": 1e6 * ts / us_staged,
           "arrays_staged": "the default contract: nothing registered, x copied into and F, G out of the library's pinned buffers",
           "us_per_call_via_python_ctypes": 1e6 * dt / calls}
    if cfg is not None:
        rec["config"] = cfg
    return rec


def two_streams_record(tol_amd, torch, args, B, device, steps=100):
    """Side record: TWO independent batches of the headline shape in flight on two HIP streams (each batch object on
    its own stream, as the stream contract asks).  The launch tail of one evaluation and the gap between launches are
    filled by the other batch's waves.  Not the headline: a step of the headline is one batch on one stream."""
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    sets = []
    for k in range(2):
        bt = tol_amd.Batch(args.mission, (args.aircraft,), ts=args.ts, dtype=args.dtype, device=device)
        bt.set_trajectories(make_trajectories(tol_amd, B, k * B, args.mission, 1))
        dXs, dF, dG = make_inputs(bt, torch, B, 77 + k, 2)
        sets.append((bt, dXs, dF, dG, torch.empty(B, dtype=dF.dtype, device=dF.device)))
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            for k, (bt, dXs, dF, dG, obj) in enumerate(sets):
                with torch.cuda.stream(streams[k]):
                    bt.eval(dXs[i % 2], dF, dG, obj=obj)
    settle(lambda i: run(1), torch.cuda.synchronize, 5)
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(s[4]).all()) for s in sets)
    assert ok
    alg = sets[0][0].algorithmic_bytes(B)
    for s in sets:
        s[0].close()
    return {"workload": f"two batches of {B} trajectories in flight on two streams", "evaluations": 2 * steps,
            "us_per_evaluation": 1e6 * wall / (2 * steps), "node_evals_per_s": 2 * steps * B * args.ts / wall,
            "algorithmic_GBs": alg * 2 * steps / wall / 1e9, "frac_of_hbm_peak": alg * 2 * steps / wall / 1e9 / HBM_PEAK_GBS,
            "frac_of_box_fill": alg * 2 * steps / wall / 1e9 / box_fill(torch, sets[0][2].device)}


def single_gpu_records(tol_amd, torch, device):
    """BASELINE.json configs that are one-GPU cases, each at its own sizes: the single-trajectory
    configs device-resident and through the callback, and the shares one of 8 GPUs gets of configs[3] and [4].
    configs[0] is the CPU/SNOPT plumbing case (tests/test_cpp_plumbing.py)."""
    recs = []
    recs.append(device_record(tol_amd, torch, 1, "configs[1] problemS10/tempest/ts=200, one trajectory", "S10", ("tempest",), 200, 1, "f64", 200, device, 1))
    recs.append(callback_mode(tol_amd, "S10", "tempest", 200, 400, cfg=1))
    recs.append(device_record(tol_amd, torch, 2, "configs[2] problemS10/skywalker/ts=2000, one trajectory", "S10", ("skywalker",), 2000, 1, "f64", 200, device, 1))
    recs.append(callback_mode(tol_amd, "S10", "skywalker", 2000, 200, cfg=2))
    recs.append(device_record(tol_amd, torch, 3, "configs[3] share of one of 8 GPUs: batch=128", "S10", ("tempest",), 200, 128, "f64", 200, device))
    for dtype in ("f64", "f32"):
        recs.append(device_record(tol_amd, torch, 4, "configs[4] share of one of 8 GPUs: batch=1024", "mixed", AIRCRAFT5, 200, 1024, dtype, 100, device))
    return recs


class Job:
    """The process group of this run (one rank per GPU) and the one collective the path has."""

    def __init__(self, torch, dist, world, rank, local, backend, collective=None):
        self.torch, self.dist, self.world, self.rank, self.local, self.backend = torch, dist, world, rank, local, backend
        # whether the collectives of the N > 1 path are issued: always with several ranks; with ONE rank only under
        # --single-rank-collectives (a rehearsal of the RCCL calls on a box with one GPU)
        self.collective = (world > 1) if collective is None else collective

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.collective:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local])
            else:
                self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device="cuda")
        if self.collective:
            if self.backend == "nccl":
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            else:
                c = t.cpu()
                self.dist.all_reduce(c, op=self.dist.ReduceOp.MAX)
                t = c
        return [float(v) for v in t]


def sharded_run(tol_amd, job, mission, aircraft, ts, dtype, pattern, per_gpu, global_batch, steps, warmup, x_buffers, instrumented=True,
                keep=None):
    """Exactly `steps` steps of one workload over the job's ranks between two barriers.  A step = this rank's shard in ONE
    launch (F, G and the objectives) + the all-gather of the objectives (N > 1; asynchronous, four buffers in rotation, so it
    overlaps the next step's launch).  per_gpu > 0: weak scaling, that many trajectories per rank; else global_batch
    trajectories split by shard_bounds (strong scaling).  Returns the timings as maxima over the ranks."""
    from tol_amd.distributed import shard_bounds
    torch, dist, world, rank = job.torch, job.dist, job.world, job.rank
    coll = job.collective
    if per_gpu > 0:
        B, first, total, scaling = per_gpu, rank * per_gpu, per_gpu * world, "weak"
        Bmax = B
    else:
        lo, hi = shard_bounds(global_batch, rank, world)
        B, first, total, scaling = hi - lo, lo, global_batch, "strong"
        Bmax = shard_bounds(total, 0, world)[1]                     # widest shard (gather buffer)
    # keep (a dict): the batch object and its buffers outlive this run under the workload's key, and a later run of the
    # SAME workload on this rank re-uses them -- the speed of a launch depends on where its output buffers landed
    # (profiles/r04_allocation_classes.md), so two records of one workload are taken on one allocation
    key = (mission, tuple(aircraft), ts, dtype, pattern, B, first, x_buffers)
    reused = keep is not None and key in keep
    if reused:
        bt, dXs, dF, dG = keep[key]
    else:
        bt = tol_amd.Batch(mission, aircraft, ts=ts, dtype=dtype, device=job.local, pattern=pattern)
        bt.set_trajectories(make_trajectories(tol_amd, max(B, 1), first, mission, len(aircraft)))
        dXs, dF, dG = make_inputs(bt, torch, max(B, 1), first, x_buffers)
        if keep is not None:
            keep[key] = (bt, dXs, dF, dG)
    # Objective buffers in rotation: the launch of step i writes obj[i % NOBJ] and must wait for the gather that last read it,
    # the one of step i - NOBJ.  With two buffers that wait was on the critical path (measured with one RCCL rank: 303 instead
    # of 277 us per step): a launch keeps every CU's LDS and wave slots booked until its last tiles, so the gather's kernel,
    # queued behind launch i-2, only gets onto the chip in the tail of launch i-1 -- and launch i then stood waiting for it.
    # Four buffers (and a high-priority stream for RCCL, main()) take the gather off that path.
    NOBJ = 4
    obj = [torch.zeros(Bmax, dtype=dF.dtype, device=dF.device) for _ in range(NOBJ)]
    gdev = dF.device if job.backend == "nccl" else torch.device("cpu")
    allobj = [torch.empty(Bmax * world, dtype=dF.dtype, device=gdev) for _ in range(NOBJ)] if coll else None
    pending = [None] * NOBJ

    def step(i):
        s = i % NOBJ
        if coll and pending[s] is not None:
            # buffer reuse: the gather of step i - NOBJ must be done.  This process runs NOBJ steps ahead of its GPU, so it is
            # normally NOT done yet: the HOST waits for it here (the launch stream still holds NOBJ - 1 launches to run
            # meanwhile) instead of putting a wait marker into the launch stream, which costs the launch behind it a few
            # microseconds -- every step (tolfg_multi does the same: profiles/r05_native_multi.md, 11.9 -> 5.8 us of gather
            # cost per step on the headline).  It also bounds how far the host runs ahead.
            w = pending[s]
            while not w.is_completed():
                pass
        if B > 0:
            bt.eval(dXs[i % len(dXs)], dF, dG, obj=obj[s], B=B)     # the finalizing waves also write the objectives, contiguous
        if coll:
            src = obj[s] if job.backend == "nccl" else obj[s].cpu()
            pending[s] = dist.all_gather_into_tensor(allobj[s], src, async_op=True)

    def fence():
        for s in range(NOBJ):
            if pending[s] is not None:
                pending[s].wait()
                pending[s] = None
        job.barrier()

    warm_run = settle(step, fence, warmup, job.max_over_ranks if coll else None)
    # The timed region: exactly `steps` steps between two barriers.  Two HIP events on the launch stream bracket the
    # launches (before the first, after the last): their distance / steps is the average time per launch, the ~2 us
    # between dependent launches included -- an upper bound of the kernel's own duration, taken without instrumenting it.
    bt.set_timing(False)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        step(i)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / steps
    last = (steps - 1) % NOBJ
    assert B == 0 or torch.isfinite(obj[last][:B]).all(), "non-finite objective"
    if coll:      # every rank's shard arrived in place, in global trajectory order
        mine = obj[last].to(allobj[last].device)
        assert torch.equal(allobj[last][rank * Bmax:(rank + 1) * Bmax], mine), "gathered objectives are out of order"
    out = {"B": B, "total": total, "scaling": scaling, "x_buffers": len(dXs), "buffers_reused": reused,
           "placement": getattr(bt, "placement", None)}
    # the gather alone (N > 1): synchronous all-gathers of the same buffers, nothing else in flight
    gather_us = 0.0
    if coll:
        reps = 50
        src = obj[0] if job.backend == "nccl" else obj[0].cpu()
        for _ in range(5):
            dist.all_gather_into_tensor(allobj[0], src)
        job.barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            dist.all_gather_into_tensor(allobj[0], src)
        torch.cuda.synchronize()
        gather_us = 1e6 * (time.perf_counter() - t1) / reps
        job.barrier()
    # Outside the timed region, the same steps again with start / stop events attached to every dispatch (the library's
    # tolfg_batch_set_timing; rocprofv3 --kernel-trace turns the same dispatch profiling on): per-launch durations, but
    # each launch then takes 10-16 us longer, part of it inside the reported duration (profiles/r02_event_cost.md).
    inst_ms = inst_min_ms = inst_step_ms = 0.0
    if instrumented:          # every rank runs the same steps (a rank with an empty shard still takes part in the gathers)
        bt.set_timing(B > 0)
        t1 = time.perf_counter()
        for i in range(steps):
            step(i)
        fence()
        inst_step_ms = 1e3 * (time.perf_counter() - t1) / steps
        if B > 0:
            nlaunch, inst_ms, inst_min_ms = bt.kernel_time()
            assert nlaunch == steps
        bt.set_timing(False)
    elapsed, kern_ms, gather_us = job.max_over_ranks([elapsed, kern_ms, gather_us])
    out.update(elapsed=elapsed, kern_ms=kern_ms, gather_us=gather_us, inst_ms=inst_ms, inst_min_ms=inst_min_ms, inst_step_ms=inst_step_ms,
               alg_bytes=bt.algorithmic_bytes(B) if B > 0 else 0.0, warmup_steps_run=warm_run)
    # Calibration of THIS box in THIS process, right after the timed region (same clocks, same temperature): the vendor's
    # fill kernel, and the bare store loop of this launch's own shape on the same buffers (rank 0's figures are reported)
    out["box_fill_GBs"] = box_fill(torch, dF.device)
    if B > 8:
        out["box_stream_shape_GBs"], out["box_stream_shape_us"] = store_shape_rate(bt, torch, dXs, dF, dG, B, ts, 46 if pattern == "compact" else 104)
    job.barrier()
    if keep is None:
        bt.close()
    del dXs, dF, dG, obj, allobj
    if keep is None:
        torch.cuda.empty_cache()
    return out


def stated_config_records(tol_amd, job, x_buffers, keep=None):
    """configs[3] and configs[4] AS STATED in BASELINE.json -- a global batch of 1024 S10 trajectories, a global mixed
    batch of 8192 in fp64 and fp32 -- sharded over however many GPUs the job has (strong scaling): per step every rank
    evaluates its shard in one launch and the objectives are all-gathered."""
    recs = []
    for cfg, what, mission, aircraft, G, dtype, steps in (
            (3, "configs[3] batch=1024 problemS10 ts=200, randomized wind/IC", "S10", ("tempest",), 1024, "f64", 200),
            (4, "configs[4] mixed G7+S10 batch=8192, five air-frames, ts=200", "mixed", AIRCRAFT5, 8192, "f64", 100),
            (4, "configs[4] mixed G7+S10 batch=8192, five air-frames, ts=200", "mixed", AIRCRAFT5, 8192, "f32", 100)):
        r = sharded_run(tol_amd, job, mission, aircraft, 200, dtype, "reference", 0, G, steps, 10, x_buffers, keep=keep)
        if job.rank != 0:
            continue
        world = job.world
        # rank 0's shard is the widest: its bytes bound every rank's; the job moves `total` trajectories per step
        alg_total = r["alg_bytes"] * G / max(r["B"], 1)
        recs.append({"config": cfg, "workload": what + (f", global batch sharded over {world} GPUs (strong scaling)" if world > 1 else ", one launch"),
                     "mode": "device-resident", "batch": G, "batch_per_gpu": r["B"], "n_gpus": world, "ts": 200, "dtype": dtype, "steps": steps,
                     "ms_per_step": 1e3 * r["elapsed"] / steps, "node_evals_per_s": G * 200 * steps / r["elapsed"],
                     "eval_us": 1e3 * r["kern_ms"], "eval_us_instrumented": 1e3 * r["inst_ms"], "eval_min_us_instrumented": 1e3 * r["inst_min_ms"],
                     "gather_us": r["gather_us"], "launches_per_step": 1,
                     "algorithmic_bytes": alg_total, "achieved_GBs": alg_total / (r["kern_ms"] * 1e-3) / 1e9,
                     "frac_of_hbm_peak": alg_total / (r["kern_ms"] * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                     "frac_of_hbm_peak_whole_step": alg_total / (r["elapsed"] / steps) / 1e9 / (HBM_PEAK_GBS * world),
                     "warmup_steps_run": r["warmup_steps_run"], "box_fill_GBs": r["box_fill_GBs"],
                     "placement": r["placement"],
                     "buffers": ("the headline's batch object and buffers (the same workload: a second, independent timed region on the "
                                 "same allocation)" if r["buffers_reused"] else "its own allocation"),
                     "frac_of_box_fill": alg_total / (r["kern_ms"] * 1e-3) / 1e9 / (r["box_fill_GBs"] * world),
                     "box_stream_shape_GBs": r.get("box_stream_shape_GBs"), "box_stream_shape_us": r.get("box_stream_shape_us"),
                     "vs_bare_store_loop": (alg_total / (r["kern_ms"] * 1e-3) / 1e9 / (r["box_stream_shape_GBs"] * world)
                                                  if r.get("box_stream_shape_GBs") else None),
                     "note": "eval_us = slowest rank's time per launch (HIP events on its launch stream); gather_us = one synchronous "
                             "all-gather of the objectives alone; fractions are of n_gpus x 8 TB/s"})
        if cfg == 3:
            # the worst point of the scaling curve explains itself: a share of this 25 MB launch is latency-bound
            recs[-1]["expected_scaling"] = {
                "per_gpu_batch": {str(n): -(-G // n) for n in (1, 2, 4, 8)},
                "launch_us_measured_on_one_gpu": {"1024": 38.1, "512": 23.8, "256": 16.0, "128": 12.4},
                "collective_us_per_step": "5-7 (one-rank RCCL rehearsals, profiles/r05_native_multi.md; 11-13 in round 4, when a wait marker still went into the launch stream every step)",
                "expect": "a single device-resident trajectory already costs 10.8-12.4 us (one wave's life), so 1024 trajectories over 8 GPUs "
                          "(128 each) read ~2-2.6 x one GPU, not 8 x: strong scaling of a 25 MB launch is latency-bound; the weak-scaling "
                          "headline (8192 per GPU, 280 us launches) is the curve to read for bandwidth"}
    return recs


def device_identity(torch, ordinal):
    """What tells one GPU from another: ordinal, name, PCI bus id (torch's device properties where they carry it, else the HIP
    runtime's hipDeviceGetPCIBusId), uuid where known."""
    if not torch.cuda.is_available():
        return {"device": None, "name": "no GPU", "pci_bus_id": None, "uuid": None, "cus": None}
    props = torch.cuda.get_device_properties(ordinal)
    bus = None
    if all(hasattr(props, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    else:
        try:
            import ctypes as C
            import tol_amd
            tol_amd.lib()
            buf = C.create_string_buffer(64)
            if tol_amd.capi._hip_runtime.hipDeviceGetPCIBusId(buf, 64, int(ordinal)) == 0:
                bus = buf.value.decode()
        except Exception:      # noqa: BLE001
            bus = None
    return {"device": int(ordinal), "name": props.name, "pci_bus_id": bus, "uuid": str(getattr(props, "uuid", "")) or None,
            "cus": getattr(props, "multi_processor_count", None)}


def rank_evidence(torch, dist, job, args, banner):
    """What the N > 1 line needs to prove which devices ran: every rank's identity card, gathered on rank 0."""
    from tol_amd.distributed import shard_bounds
    card = dict(device_identity(torch, torch.cuda.current_device() if torch.cuda.is_available() else 0), rank=job.rank, local_rank=job.local, pid=os.getpid(),
                host=os.uname().nodename)
    if args.global_batch > 0:
        card["shard"] = list(shard_bounds(args.global_batch, job.rank, job.world))
    else:
        card["shard"] = [job.rank * args.batch, (job.rank + 1) * args.batch]
    cards = [card]
    if job.collective:
        cards = [None] * job.world
        dist.all_gather_object(cards, card)
    seen = dist.get_world_size() if job.collective else 1
    rccl = {"torch_nccl_version": None, "banner": banner or None}
    try:
        rccl["torch_nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:      # noqa: BLE001
        pass
    return cards, seen, rccl, rank_problems(cards, seen, args.gpus, job.backend if job.collective else None)


def rank_problems(cards, seen, gpus, backend):
    """What makes a line unreportable: a process group of another size than --gpus; under RCCL, two ranks on one device (a
    scaling point over fewer GPUs than it says); ranks missing or out of order."""
    problems = []
    if seen != gpus:
        problems.append(f"the process group has {seen} ranks, --gpus says {gpus}")
    if [c.get("rank") for c in cards] != list(range(len(cards))) or len(cards) != seen:
        problems.append(f"identity cards of ranks {[c.get('rank') for c in cards]} for a group of {seen}")
    if backend == "nccl":
        ids = [(c.get("host"), c.get("pci_bus_id") or ("ordinal %s" % c.get("device"))) for c in cards]
        if len(set(ids)) != len(ids):
            problems.append(f"the ranks do not sit on pairwise distinct devices: {ids}")
    return problems


def capture_stdout_begin():
    """RCCL prints its version banner on fd 1 when the first communicator comes up: catch it in a file (fd 1 is pointed at stderr
    otherwise, main())."""
    import tempfile
    tmp = tempfile.TemporaryFile()
    os.dup2(tmp.fileno(), 1)
    return tmp


def capture_stdout_end(tmp):
    sys.stdout.flush()
    os.dup2(2, 1)
    tmp.seek(0)
    text = tmp.read().decode(errors="replace")
    tmp.close()
    if text:
        sys.stderr.write(text)
    keep = [ln.strip() for ln in text.splitlines() if "RCCL" in ln or "NCCL" in ln or "HIP version" in ln or "ROCm version" in ln]
    return " | ".join(keep)[:600]


# ------------------------------------------------------------------------------------ the native C++ host path

class _Raw:
    """A device pointer as something torch can view (no ownership)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def native_workload(tol_amd, torch, devices, mission, aircraft, ts, dtype, total, steps, warmup, x_buffers, issue, scaling, what, gather="rccl"):
    """One workload through the native several-GPUs-one-process host path (tol_amd/csrc/multi.cpp: one launch stream, one gather
    stream and one issuing thread per device, grouped ncclAllGather of the objectives over four rotating buffers), the step loop
    itself issued from native code (tolfg_multi_time_steps).  Inputs as bench.py's torch.distributed path makes them: initial
    guesses generated on the device + 5 % noise, x_buffers input sets in rotation.  torch is used for that noise only."""
    nd = len(devices)
    m = tol_amd.Multi(mission, aircraft, ts=ts, dtype=dtype, devices=devices)
    m.set_issue(issue)
    m.set_gather(gather)
    t0 = time.perf_counter()
    m.set_trajectories(make_trajectories(tol_amd, total, 0, mission, len(aircraft)))
    setup_s = time.perf_counter() - t0
    m.x0()
    m.sync()
    typestr = "<f8" if dtype == "f64" else "<f4"
    sets = [[] for _ in range(max(x_buffers, 1))]
    keep, shards = [], []
    for i, d in enumerate(devices):
        lo, hi = m.shard(i)
        shards.append([lo, hi])
        (dX, ldx), _, _ = m.buffers(i)
        rows = max(hi - lo, 1)
        with torch.cuda.device(d):
            base = torch.as_tensor(_Raw(dX, (rows, ldx), typestr), device=torch.device("cuda", d))
            assert base.data_ptr() == dX, "torch did not view the library's X in place"
            gen = torch.Generator(device=base.device).manual_seed(5000000 + lo)
            noise = torch.rand(base.shape, dtype=base.dtype, device=base.device, generator=gen) * 2 - 1
            dt = base[:, 0].clone()
            base += 0.05 * noise * (1 + base.abs())
            base[:, 0] = dt.abs() + 0.01
            sets[0].append(dX)
            for j in range(1, len(sets)):
                x = torch.roll(base, shifts=10 * j, dims=0).contiguous()
                x[:, 0] = base[:, 0]
                keep.append(x)
                sets[j].append(x.data_ptr())
            torch.cuda.synchronize(d)
    # warm-up: at least `warmup` steps and MIN_WARM_S of launching
    probe = m.time_steps(5, warm=2, x_sets=sets)
    warm = max(int(warmup), int(math.ceil(MIN_WARM_S / max(probe["wall_us_per_step"] * 1e-6, 1e-7))))
    t = m.time_steps(steps, warm=min(warm, 100000), x_sets=sets)
    plain = m.time_steps(steps, warm=5, x_sets=sets, gather=False)       # the launches alone, for the cost of the gather in the step
    obj = m.gather_wait(m.step(dX=sets[0]))
    assert np.isfinite(obj).all() and obj.shape == (total,), "non-finite objective"
    m.sync()
    # the same object and buffers, the objectives stored straight into one pinned host vector by the finalizing waves instead of
    # all-gathered (tolfg_multi_set_gather HOST): what a host-side consumer of the objectives pays per step
    other = None
    if gather == "rccl":
        m.set_gather("host")
        h = m.time_steps(steps, warm=max(5, min(warm, 2000)), x_sets=sets)
        obj2 = m.gather_wait(m.step(dX=sets[0]))
        assert np.array_equal(obj2, obj), "the host-gathered objectives differ from the all-gathered ones"
        other = {"gather": "host", "ms_per_step": 1e-3 * h["wall_us_per_step"], "eval_us": h["launch_us_per_step"], "gather_us": h["gather_us"],
                 "issue_us_per_step": h["issue_us_per_step"], "node_evals_per_s": total * ts / (h["wall_us_per_step"] * 1e-6)}
        m.set_gather("rccl")
    wall = t["wall_us_per_step"] * 1e-6
    per_dev = t["launch_us_per_device"]
    # algorithmic bytes of the widest shard through a host-side batch object of the same description (no GPU work)
    bt = tol_amd.Batch(mission, aircraft, ts=ts, dtype=dtype, device=devices[0])
    bt.set_trajectories(make_trajectories(tol_amd, shards[0][1] - shards[0][0], 0, mission, len(aircraft)))
    alg0 = bt.algorithmic_bytes(shards[0][1] - shards[0][0])
    bt.close()
    rec = {"workload": what, "scaling": scaling, "n_gpus": nd, "batch": total, "batch_per_gpu": shards[0][1] - shards[0][0], "ts": ts, "dtype": dtype,
           "steps": steps, "warmup_steps_run": warm + 2 + 5, "x_buffers": len(sets), "issue": t["issue"], "gather": t["gather"],
           "objectives_stored_to_host_instead": other,
           "ms_per_step": 1e3 * wall, "node_evals_per_s": total * ts / wall,
           "eval_us": t["launch_us_per_step"], "eval_us_per_device": per_dev, "gather_us": t["gather_us"],
           "issue_us_per_step": t["issue_us_per_step"],
           "ms_per_step_without_gather": 1e-3 * plain["wall_us_per_step"], "eval_us_without_gather": plain["launch_us_per_step"],
           "algorithmic_bytes_widest_shard": alg0,
           "frac_of_hbm_peak": (alg0 / (t["launch_us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if t["launch_us_per_step"] > 0 else None,
           "shards": shards, "setup_s": setup_s,
           "note": "step loop issued from native code (tolfg_multi_time_steps): one launch per device + the asynchronous all-gather of the "
                   "objectives on the devices' gather streams; eval_us = the slowest device's (event after its last launch - event before its "
                   "first) / steps; gather_us = one synchronous gather alone; frac_of_hbm_peak = the widest shard's bytes over eval_us"}
    del keep
    m.close()
    torch.cuda.empty_cache()
    return rec


def native_multi_main(args):
    """`bench.py --native-multi N`: ONE process, tolfg_multi over devices 0..N-1 -- the path BASELINE's north star words ("host code
    stays C++ ... RCCL over xGMI only for the final objective gather").  Prints one JSON record (merged into the bench line as
    `native_multi` by the process that started this one)."""
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import tol_amd
    n = args.native_multi
    ndev = torch.cuda.device_count()
    out = {"host_path": "native C++: tolfg_multi (tol_amd/csrc/multi.cpp), one process, one issuing thread + launch stream + gather stream per device",
           "n_gpus": n, "devices_visible": ndev}
    try:
        if ndev < n:
            raise RuntimeError(f"{n} devices asked for, {ndev} visible")
        devices = list(range(n))
        cards = [device_identity(torch, d) for d in devices]
        ids = [c["pci_bus_id"] or ("ordinal %d" % c["device"]) for c in cards]
        if len(set(ids)) != len(ids):
            raise RuntimeError(f"the devices are not pairwise distinct: {ids}")
        out["devices"] = cards
        tmp = capture_stdout_begin()
        try:
            aircraft = AIRCRAFT5 if args.mission == "mixed" else (args.aircraft,)
            total = args.global_batch if args.global_batch > 0 else args.batch * n
            head = native_workload(tol_amd, torch, devices, args.mission, aircraft, args.ts, args.dtype, total, args.steps, args.warmup,
                                   args.x_buffers, args.native_issue, "strong" if args.global_batch > 0 else "weak",
                                   f"the headline: {args.mission} batch, ts={args.ts}, {args.dtype}, {total} trajectories over {n} device(s)")
        finally:
            banner = capture_stdout_end(tmp)
        out.update(head)
        out["value"] = head["node_evals_per_s"]
        out["unit"] = "node-evals/s"
        L = tol_amd.lib()
        out["rccl"] = {"library": L.tolfg_multi_rccl_library().decode(), "version_code": L.tolfg_multi_rccl_version(), "banner": banner or None}
        if not args.no_configs:
            recs = []
            for cfg, what, mission, air, G, dtype, steps in (
                    (3, "configs[3] batch=1024 problemS10 ts=200, randomized wind/IC", "S10", ("tempest",), 1024, "f64", 200),
                    (4, "configs[4] mixed G7+S10 batch=8192, five air-frames, ts=200", "mixed", AIRCRAFT5, 8192, "f64", 100),
                    (4, "configs[4] mixed G7+S10 batch=8192, five air-frames, ts=200", "mixed", AIRCRAFT5, 8192, "f32", 100)):
                r = native_workload(tol_amd, torch, devices, mission, air, 200, dtype, G, steps, 10, args.x_buffers, args.native_issue, "strong",
                                    what + f", global batch sharded over {n} device(s)")
                r["config"] = cfg
                recs.append(r)
            out["configs"] = recs
    except Exception as exc:      # noqa: BLE001
        import traceback
        traceback.print_exc()
        out["error"] = f"{type(exc).__name__}: {exc}"
    data = (json.dumps(out) + "\n").encode()
    while data:
        data = data[os.write(result_fd, data):]
    return 0 if "error" not in out else 1


def run_native_child(n, args, issue="grouped", timeout=420):
    """Start `bench.py --native-multi n` as a CHILD process (never an exec: this process has initialised the GPU) and return its
    record, or a record that says why there is none."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--native-multi", str(n), "--native-issue", issue, "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--batch", str(args.batch), "--ts", str(args.ts), "--mission", args.mission,
           "--aircraft", args.aircraft, "--dtype", args.dtype, "--x-buffers", str(args.x_buffers), "--min-warm-seconds", str(MIN_WARM_S)]
    if args.global_batch > 0:
        cmd += ["--global-batch", str(args.global_batch)]
    if args.no_configs:
        cmd.append("--no-configs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                           "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "OMP_NUM_THREADS")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"the native leg did not finish within {timeout} s", "n_gpus": n, "issue": issue}
    rec = None
    for ln in res.stdout.splitlines():
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            try:
                rec = json.loads(ln)
            except ValueError:
                rec = None
    if rec is None:
        rec = {"error": f"the native leg ended with code {res.returncode} and no record", "n_gpus": n, "issue": issue}
    rec["child_wall_s"] = time.perf_counter() - t0
    rec["child_exit_code"] = res.returncode
    return rec


def wait_for_pids(pids, seconds):
    """Rank 0 waits (bounded) for the other ranks' processes to be gone, so that the native leg finds the GPUs idle."""
    end = time.perf_counter() + seconds
    left = [p for p in pids if p != os.getpid()]
    while left and time.perf_counter() < end:
        alive = []
        for p in left:
            try:
                with open(f"/proc/{p}/stat") as fh:              # "pid (comm) state ...": a zombie has let go of its GPU already
                    state = fh.read().rsplit(")", 1)[1].split()[0]
                if state != "Z":
                    alive.append(p)
            except (OSError, IndexError):
                pass
        left = alive
        if left:
            time.sleep(0.05)
    return left


def spawn_ranks(n):
    """`python bench.py --gpus N` (N > 1) started as a plain command: start the N ranks as a child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same arguments>`, one rank per GPU,
    rendezvous on 127.0.0.1 at a free port.  Called before torch is imported or any GPU call is made.  The child's
    stderr passes through; of its stdout, rank 0's JSON line is printed last, on its own line, and anything else goes
    to stderr.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in child.stdout:
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks ended without a result line\n")
        rc = 1
    return rc


def main():
    global MIN_WARM_S, CALIBRATE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=8192, help="trajectories per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="total trajectories over all GPUs instead of --batch per GPU (strong scaling)")
    ap.add_argument("--ts", type=int, default=200)
    ap.add_argument("--mission", default="mixed", choices=["S10", "G7", "mixed"])
    ap.add_argument("--aircraft", default="tempest", help="air-frame of a single-mission batch; a mixed batch uses all five")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-BASELINE-config records and side runs")
    ap.add_argument("--x-buffers", type=int, default=4,
                    help="distinct input batches rotated over the steps; 4 x 145 MB exceed the 256 MiB Infinity "
                         "Cache, so every step reads its X from HBM rather than from a cache that kept it")
    ap.add_argument("--pattern", default="reference", choices=["reference", "compact"],
                    help="Jacobian sparsity pattern; the headline metric is quoted on the reference's own pattern")
    ap.add_argument("--min-warm-seconds", type=float, default=MIN_WARM_S,
                    help="every timed region is preceded by at least this much back-to-back launching (and >= --warmup steps)")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip the same-process box calibration (vendor fill, bare store loop); for profiler passes")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1: nccl = RCCL over xGMI; gloo (objectives staged through host "
                         "memory) only rehearses the multi-rank step loop, e.g. several ranks sharing one GPU")
    ap.add_argument("--single-rank-collectives", action="store_true",
                    help="rehearsal on a box with one GPU: a process group of ONE rank on --backend, and every collective of the "
                         "N > 1 path is issued (barrier, all-reduce of the timings, the asynchronous all-gather of the objectives)")
    ap.add_argument("--native-multi", type=int, default=0,
                    help="ONE process over devices 0..N-1 through the native C++ host path (tolfg_multi_*); prints that leg's record. "
                         "A normal run starts this as a child after its own measurements and merges the record as `native_multi`")
    ap.add_argument("--native-issue", default="grouped", choices=["grouped", "threads"],
                    help="how the native leg issues the all-gather: one ncclGroupStart/End bracket | one call per device thread")
    ap.add_argument("--no-native-multi", action="store_true", help="skip the native C++ leg")
    args = ap.parse_args()

    MIN_WARM_S, CALIBRATE = max(args.min_warm_seconds, 0.0), not args.no_calibration
    if args.native_multi > 0:
        sys.exit(native_multi_main(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain command: nothing has touched a GPU yet, so the ranks are started from here as a CHILD
        # (never an exec) and this process only relays rank 0's line and the child's exit code
        sys.exit(spawn_ranks(args.gpus))

    # This process's stdout carries ONE line, the result.  Libraries write there too -- RCCL prints a version banner
    # ("RCCL version : ...", four lines) on stdout when its first communicator comes up -- so file descriptor 1 is pointed
    # at stderr for everything else, and the result line goes out through a private copy of the original descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import tol_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a line "
                         f"under the wrong n_gpus\n")
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local = local % max(ndev, 1)               # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    coll = world > 1 or args.single_rank_collectives
    banner = ""
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:       # started as a plain command: the rendezvous of a world of one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            # eager communicator creation on this rank's GPU (RCCL over xGMI) and a first collective outside any
            # timing (RCCL builds its rings lazily).  A failure here is fatal: a scaling point measured over a
            # host-staged fallback would not be "RCCL over xGMI"; gloo is only ever chosen with --backend gloo.
            banner_file = capture_stdout_begin()
            try:
                # RCCL's kernels on a high-priority stream: a collective queued behind a launch that has every CU booked gets
                # onto the chip when the first wave slots free up, not when the launch's backlog of tiles is through
                opts = None
                try:
                    opts = dist.ProcessGroupNCCL.Options()
                    opts.is_high_priority_stream = True
                except Exception:      # noqa: BLE001
                    opts = None
                import datetime
                limit = datetime.timedelta(seconds=300)      # a collective that never completes ends the run with an error, not a hang
                try:
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=opts, timeout=limit)
                except TypeError:
                    dist.init_process_group("nccl", timeout=limit)
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
                banner = capture_stdout_end(banner_file)
            except Exception as exc:      # noqa: BLE001
                os.dup2(2, 1)
                sys.stderr.write(f"bench.py rank {rank}: RCCL is not usable here: {type(exc).__name__}: {exc}\n"
                                 f"  (no fallback is taken; rerun with --backend gloo only to rehearse the step loop)\n")
                sys.stderr.flush()
                os._exit(3)
        else:
            dist.init_process_group("gloo")
    job = Job(torch, dist, world, rank, local, args.backend, collective=coll)
    # which devices run this line: every rank's identity card on rank 0; a rank on a device another rank already has, or a
    # process group of another size than --gpus, ends the run (a scaling point must not be reported over the wrong hardware)
    cards, world_seen, rccl_info, problems = rank_evidence(torch, dist, job, args, banner)
    if problems:
        if rank == 0:
            sys.stderr.write("bench.py: " + "; ".join(problems) + "\n")
        if coll:
            dist.destroy_process_group()
        sys.exit(4)

    aircraft = AIRCRAFT5 if args.mission == "mixed" else (args.aircraft,)
    keep = {}        # the headline's batch and buffers: re-used by a stated config of the same workload (one GPU: configs[4] fp64)
    r = sharded_run(tol_amd, job, args.mission, aircraft, args.ts, args.dtype, args.pattern,
                    0 if args.global_batch > 0 else args.batch, args.global_batch, args.steps, args.warmup, args.x_buffers, keep=keep)
    # where every rank's output buffer landed (the placement search of its G buffer): onto its identity card
    placements = [r["placement"]]
    if coll:
        placements = [None] * world
        dist.all_gather_object(placements, r["placement"])
    for card, pl in zip(cards, placements):
        card["placement"] = {"candidates": pl.get("candidates"), "probe_us": pl.get("probe_us")} if pl else None
    configs = None
    if not args.no_configs:
        configs = stated_config_records(tol_amd, job, args.x_buffers, keep=keep)      # every rank takes part
    x_cached = None
    if world == 1 and not args.no_configs and keep:
        x_cached = same_x_record(torch, next(iter(keep.values())), r["B"], args.ts, min(args.steps, 100))
    for held in keep.values():
        held[0].close()
    keep.clear()
    torch.cuda.empty_cache()

    if rank == 0:
        B, total = r["B"], r["total"]
        value = total * args.ts * args.steps / r["elapsed"]
        alg_bytes = r["alg_bytes"]
        achieved = alg_bytes / (r["kern_ms"] * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    tj = json.load(fh)
                if (tj.get("batch") == B and tj.get("ts") == args.ts and tj.get("dtype") == args.dtype
                        and tj.get("mission", "S10") == args.mission and tj.get("pattern", "reference") == args.pattern):
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/traffic_latest.json (replayed: PMC passes of " + str(tj.get("source", "tools/profile_gpu.sh")) + ", not measured in this run)"
            except (OSError, ValueError):
                traffic = None
        backend = "none (single GPU)" if not coll else ("rccl" if args.backend == "nccl" else "gloo (host-staged rehearsal)")
        if coll and world == 1:
            backend += ", ONE rank: a rehearsal of the collectives (--single-rank-collectives)"
        if args.mission == "mixed":
            what = (f"BASELINE configs[4]: mixed problemG7 + problemS10 batch (mission = b mod 2), all five aircraft .param files "
                    f"(b mod 5), ts={args.ts}, {args.dtype}")
        else:
            what = f"problem{args.mission}/{'+'.join(aircraft)}.param/ts={args.ts}, {args.dtype} (BASELINE configs[1] batched as configs[3] prescribes)"
        line = {
            "metric": "collocation-node F/G+Jacobian evals/sec",
            "value": value, "unit": "node-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_steps_run": r["warmup_steps_run"],
            "ms_per_step": 1e3 * r["elapsed"] / args.steps, "higher_is_better": True, "scaling": r["scaling"],
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "backend": backend,
            "config": {"workload": what + f": a device-resident batch of {B} trajectories per GPU with randomized shear wind and "
                                          f"start offsets; one F+G launch + objective gather per step",
                       "mission": args.mission, "aircraft": "+".join(aircraft), "ts": args.ts, "pattern": args.pattern,
                       "batch_per_gpu": B, "global_batch": total, "x_buffers": r["x_buffers"],
                       "output_placement": r["placement"],
                       "parallelism": (f"batch-sharded x{world}, {backend} all-gather of objectives") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "box_fill_GBs": r["box_fill_GBs"], "frac_of_box_fill": achieved / r["box_fill_GBs"],
                         "box_stream_shape_GBs": r.get("box_stream_shape_GBs"), "box_stream_shape_us": r.get("box_stream_shape_us"),
                         "vs_bare_store_loop": achieved / r["box_stream_shape_GBs"] if r.get("box_stream_shape_GBs") else None,
                         "calibration": "reference points, not ceilings, measured in this process on this box right after the timed region: "
                                        "box_fill = the vendor's fill kernel (torch fill_) over 800 MB; box_stream_shape = this launch's own "
                                        "grid, tile order, resident-wave cap and store flavour with nothing in it but the Jacobian-slab stores "
                                        "(tolfg_batch_set_store_shape), GB/s over the bytes it stores; frac_of_box_fill / vs_bare_store_loop = "
                                        "achieved / that (the latter may exceed 1)",
                         "kernel": "tolfg::fg_kernel (the whole evaluation: the last tile wave of a trajectory finalizes it)",
                         "kernel_ms": r["kern_ms"],
                         "timing": "kernel_ms = (HIP event after the last launch - HIP event before the first) / steps, on the launch "
                                   "stream, over the timed region: it contains the gap between dependent launches, so the kernel "
                                   "itself is no slower than this. instrumented_*: a second pass with start/stop events attached "
                                   "to every dispatch, as rocprofv3 --kernel-trace does; dispatch profiling itself lengthens every "
                                   "launch (profiles/r02_event_cost.md)",
                         "instrumented_kernel_ms": r["inst_ms"], "instrumented_kernel_min_ms": r["inst_min_ms"], "instrumented_ms_per_step": r["inst_step_ms"],
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_node": alg_bytes / max(B * args.ts, 1)},
        }
        line["world_seen"] = world_seen
        line["ranks"] = cards
        if coll:
            line["gather_us"] = r["gather_us"]
            line["rccl"] = rccl_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        if configs is not None:
            line["configs"] = configs
            if world == 1:
                line["configs"] = single_gpu_records(tol_amd, torch, local) + configs
                line["configs"].sort(key=lambda c: (c["config"], c["mode"] != "device-resident", -c.get("batch", 0)))
                line["callback"] = [c for c in line["configs"] if c["mode"] == "callback"]
                # side records on the round-2 headline shape (S10 / tempest / 4096): not BASELINE configs
                side = argparse.Namespace(mission="S10", aircraft="tempest", ts=200, dtype="f64", x_buffers=args.x_buffers)
                s4096 = device_record(tol_amd, torch, 9, "side: problemS10/tempest/ts=200, 4096 trajectories (the round-1/2 headline shape)",
                                      "S10", ("tempest",), 200, 4096, "f64", 100, local)
                del s4096["config"]
                line["s10_batch_4096"] = s4096
                line["next_compact_pattern"] = compact_side_run(tol_amd, torch, side, 4096, local)
                line["two_batches_two_streams"] = two_streams_record(tol_amd, torch, side, 4096, local)
                if x_cached is not None:
                    line["headline_same_x_every_step"] = x_cached
    if coll:
        job.barrier()
        dist.destroy_process_group()

    if rank == 0:
        # The native C++ host path (tolfg_multi: one process over all the devices), measured in a CHILD process after this one's
        # work is done and -- under a launcher -- after the other ranks have gone (they exit right after the barrier above).
        if not args.no_native_multi:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            left = wait_for_pids([c["pid"] for c in cards], 30.0) if world > 1 else []
            # the native leg takes the devices this process can see: all `world` of them under RCCL (one rank per GPU), fewer when a
            # gloo rehearsal shared GPUs or the launcher masked the devices per rank
            n_native = min(world, max(ndev, 1))
            # Over several devices the collective is issued by one thread per device (tolfg_multi ISSUE_THREADS): the group bracket is
            # serial work of ~18 us per device for one thread -- 150 us per step at eight devices, more than configs[3]'s and configs[4]'s
            # shards take to run (profiles/r05_native_multi.md, "What a step costs the HOST").  The bracket form (the library's default)
            # is measured too, in a process of its own, headline and configs.
            first = "threads" if n_native > 1 else "grouped"
            nm = run_native_child(n_native, args, first)
            if left:
                nm["ranks_still_alive_at_start"] = left
            if n_native != world:
                nm["note_devices"] = f"{world} ranks, {ndev} device(s) visible to rank 0: the native leg ran over {n_native}"
            line["native_multi"] = nm
            if n_native > 1 and "error" not in nm:
                line["native_multi_grouped"] = run_native_child(n_native, args, "grouped", timeout=300)
        sys.stdout.flush()
        data = (json.dumps(line) + "\n").encode()
        while data:
            data = data[os.write(result_fd, data):]


def same_x_record(torch, held, B, ts, steps):
    """Side record, not the headline: the headline's batch on the headline's buffers with the SAME X buffer every step, so that
    the x windows (145 MB at B = 8192, fp64) come from the 256 MiB Infinity Cache instead of HBM -- what an evaluation sees
    whose x was written just before it.  The headline itself rotates x_buffers X buffers so that x comes from HBM
    (BASELINE: inputs resident in HBM).  profiles/r04_allocation_classes.md: with x in the cache the evaluation hardly
    depends on the placement class of the output buffer."""
    bt, dXs, dF, dG = held
    obj = torch.empty(B, dtype=dF.dtype, device=dF.device)
    wall, kms, _, _ = timed_evals(bt, torch, dXs[:1], dF, dG, B, steps, 5, obj)
    alg = bt.algorithmic_bytes(B)
    gbs = alg / (kms * 1e-3) / 1e9
    return {"workload": "the headline's batch and buffers, the same X buffer every step (x from the Infinity Cache)", "steps": steps,
            "kernel_ms": kms, "node_evals_per_s": B * ts * steps / wall, "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
            "note": "frac over the same algorithmic bytes as the headline, of which the x bytes did not come from HBM here"}


def compact_side_run(tol_amd, torch, args, B, device, steps=50):
    """SURVEY.md section 8(f) rank 1, measured beside the headline: the same batch through the compact
    sparsity pattern (46 instead of 104 entries per node).  Not the headline metric -- it changes
    the pattern handed to SNOPT."""
    bc = tol_amd.Batch(args.mission, (args.aircraft,), ts=args.ts, dtype=args.dtype, device=device, pattern="compact")
    bc.set_trajectories(make_trajectories(tol_amd, B, 0, args.mission, 1))
    dXs, dF, dG = make_inputs(bc, torch, B, 0, args.x_buffers)
    obj = torch.empty(B, dtype=dF.dtype, device=dF.device)
    wall, kms, _, _ = timed_evals(bc, torch, dXs, dF, dG, B, steps, 5, obj)
    alg = bc.algorithmic_bytes(B)
    gbs = alg / (kms * 1e-3) / 1e9
    shape_gbs, shape_us = store_shape_rate(bc, torch, dXs, dF, dG, B, args.ts, 46)
    return {"value": B * args.ts * steps / wall, "unit": "node-evals/s", "steps": steps, "ms_per_step": 1e3 * wall / steps,
            "kernel_ms": kms, "algorithmic_bytes_per_launch": alg, "achieved_GBs": gbs,
            "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "bytes_per_node": alg / (B * args.ts),
            "frac_of_box_fill": gbs / box_fill(torch, dF.device), "box_stream_shape_GBs": shape_gbs, "box_stream_shape_us": shape_us,
            "vs_bare_store_loop": gbs / shape_gbs}


if __name__ == "__main__":
    main()
