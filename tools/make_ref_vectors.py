#!/usr/bin/env python3
"""BUILD-CONTAINER TOOL: reference-derived golden vectors for the user-function path.

Reads the reference's sources under /root/reference at RUN TIME (they are never copied), evaluates the
functions on the path with tools/refeval/cinterp.py -- a small interpreter for the C++ subset they are
written in -- and stores only NUMBERS (inputs and the outputs the reference's own code produced) in
tests/golden/ref_eval_vectors.npz.  Only that .npz travels to the GPU box.

What is evaluated, per case, exactly in the reference's own calling order
(src/problemS10.cpp:9-17 / src/problemG7.cpp:9-17 constructors, src/DefineFG.cpp:9-48):
    InitialCond()  -> x0            setLimits() -> xlow, xupp, Flow, Fupp       countG(x0) -> pattern
    for every test point x:   modelWind(x);  computeF(x, F);  computeG(x, G)
with wind model 1 (the offline fallback), model 99 (a seeded per-node ENU wind table left untouched
by modelWind's `default:` arm, SURVEY.md section 8c "wind injection") and model 3 (trilinear grid,
a synthetic cache), non-shipped gains (kT, kp, kv, kdt all different and non-zero), perturbed
air-frame coefficients and seeded decision vectors.  The object state a constructor would have set
(src/problem.cpp:13-192) is supplied as data below, each member with the line it mirrors.

The interpreter reads a scalar declared without an initialiser as NaN, so the reference's 11
undefined S10 entries (src/problemS10.cpp:397) come out as NaN in G; the fixture keeps them as NaN and
the tests mask them.

usage: python tools/make_ref_vectors.py                   (writes tests/golden/ref_eval_vectors.npz, ~1 min)
       python tools/make_ref_vectors.py --set shipped     (writes tests/golden/ref_eval_shipped.npz: the repository as
                                                           shipped -- both missions x all five air frames at ts = 100, and
                                                           ts = 200 once per mission -- countG included; ~9 min on 6 cores)
       python tools/make_ref_vectors.py --set long        (writes tests/golden/ref_eval_long.npz: S10 / skywalker / ts = 2000)
       python tools/make_ref_vectors.py --set json-keys   (writes tests/golden/results_json_keys.json: the key names the reference's
                                                           result writer assigns, src/problem.cpp writeJSON)
"""
import math
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refeval"))
import cinterp  # noqa: E402

REF = "/root/reference"
OUT = os.environ.get("REF_VECTORS_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_eval_vectors.npz")
OUT_LONG = os.environ.get("REF_VECTORS_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_eval_long.npz")
OUT_SHIPPED = os.environ.get("REF_VECTORS_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_eval_shipped.npz")


def read_param_values(path):
    """Numbers of a .param file (text before the first '/', leading float) -- plain data input."""
    vals = []
    with open(path, errors="replace") as fh:
        for line in fh:
            head = line.split("/")[0].strip()
            try:
                vals.append(float(head.split("\\")[0]))
            except ValueError:
                continue
    return vals


def aircraft_members(v):
    """aircraft::aircraft, src/parameters.cpp:42-67: 15 values, three of them degrees."""
    return NS(mm=v[0], b=v[1], SS=v[2], ee=v[3], AR=v[4], Cd0=v[5], CLmin=v[6], CLmax=v[7], phimax=v[8] * math.pi / 180.0,
              Vamin=v[9], Vamax=v[10], gammamax=v[11] * math.pi / 180.0, phidotmax=v[12] * math.pi / 180.0, Tmin=v[13], Tmax=v[14])


class RefProblem:
    """A problemS10 / problemG7 object built the way the reference builds it: every data member the class declarations
    hold (include/problem.h, include/problemG7.h: read at run time) starts at its C++ default -- NaN for an
    uninitialised double -- and the reference's own constructors run on it (src/problem.cpp:13-192, then the mission's,
    src/problemS10.cpp:9-13 / src/problemG7.cpp:9-13: setLimits(); InitialCond(); countG(x)).  The four parameter
    members (ac, gn, lm, sn) are the numbers of the .param files; the database connection the constructor attempts is
    not part of the interpreted sources, so the reference's own catch block selects wind model 1, as it does offline.
    `start` other than (0, 0, 0) is a test hook: the reference hard-codes the start (src/problem.cpp:83-85), the hook
    moves it between the base constructor and the mission constructor's body."""

    def __init__(self, it, mission, N, ac, gains, lim, goal, start, derived_body=True):
        cls = "problem" + mission
        self._classes = [cls, "problem"]
        for header, c in (("problem.h", "problem"), (cls + ".h", cls)):
            for name, val in cinterp.declare_members(open(os.path.join(REF, "include", header), errors="replace").read(), c).items():
                setattr(self, name, val)
        nb = 11 if mission == "S10" else 12
        # the .param members: values only (src/parameters.cpp:42-148 reads them from the files)
        self.sn = NS(ts=int(N), numinp=11, numstates=8, numbounds=nb, opt_tol=1e-6, feas_tol=1e-6)
        self.ac = ac
        self.gn = NS(kT=gains[0], kp=gains[1], kv=gains[2], ka=gains[3], kdt=gains[4])
        self.lm = NS(dtmin=lim[0], dtmax=lim[1], xmin=lim[2], xmax=lim[3], ymin=lim[4], ymax=lim[5], zmin=lim[6], zmax=lim[7])
        east_goal, north_goal, up_goal, radius = goal
        # what arguments::arguments(char**) unpacks from `0 0 100 <east_goal> <north_goal> <up_goal> <radius> <aircraft> <mission>`
        # (src/arguments.cpp:36-46)
        args = NS(east=0.0, north=0.0, up=100.0, east_goal=float(east_goal), north_goal=float(north_goal), up_goal=float(up_goal),
                  radius_goal=float(radius), aircraft="fixture", mission=mission, root_path="./", stitch_previous=0)
        start = tuple(float(v) for v in start)

        def move_start(o):
            if start != (0.0, 0.0, 0.0):
                o.xi, o.yi, o.zi = start
        if derived_body:
            it.construct(self, cls, args, between=move_start)        # problem::problem, [hook], then setLimits / InitialCond / countG
        else:
            it.construct(self, "problem", args)                      # the "long" set: countG's probe is out of reach at ts = 2000
            move_start(self)
        assert self.Pwindmodel == 1                                   # the reference's catch block ran (src/problem.cpp:73-78)

    def set_wind_table(self, table):
        names = ("u", "v", "w", "du_dx", "du_dy", "du_dz", "dv_dx", "dv_dy", "dv_dz", "dw_dx", "dw_dy", "dw_dz")
        for f, name in enumerate(names):                                        # include/problem.h:103
            setattr(self, name, [float(t) for t in table[f]])
        self.Pwindmodel = 99                                                    # `default: break`, :732-735

    def set_grid(self, v, origin, spacing, datum):
        """cache[xi][yi][zi] = winddoc(x, y, z, u, v, w) on a regular ENU grid (include/problem.h:59-70)."""
        nx, ny, nz = v.shape
        self.cache = [[[NS(x=origin[0] + i * spacing[0], y=origin[1] + j * spacing[1], z=origin[2] + k * spacing[2],
                           u=0.0, v=float(v[i, j, k]), w=0.0) for k in range(nz)] for j in range(ny)] for i in range(nx)]
        # cacheWind builds cache[east index][north index][up index] with cache_east x cache_north x cache_up nodes
        # (src/problem.cpp:437-441).  modelWind's search loops bound the FIRST index by cache_north and the second by
        # cache_east (src/problem.cpp:556-566): on a cubic grid that makes no difference; on a non-cubic one the loops
        # still stop in the right cell for every point inside the grid (a loop that runs out of its bound ends on the
        # index it would have stopped at), and points outside it index beyond the vectors (undefined in the reference)
        self.cache_east, self.cache_north, self.cache_up = nx, ny, nz
        self.xspacing, self.yspacing, self.zspacing = (float(s) for s in spacing)
        self.EastFromDatum, self.NorthFromDatum, self.UpFromDatum = (float(d) for d in datum)
        self.Pwindmodel = 3


def perturbed(x0, N, rng, scale=0.05):
    """SURVEY.md section 8(c) recipe: every variable moved, z, Va, T put where wind and drag terms matter."""
    x = np.array(x0) + scale * rng.uniform(-1, 1, len(x0)) * (1 + np.abs(x0))
    node = x[1:].reshape(N + 1, 11)
    node[:, 2] = rng.uniform(-70, -30, N + 1)
    node[:, 3] = rng.uniform(12, 18, N + 1)
    node[:, 10] = rng.uniform(5, 15, N + 1)
    x[0] = abs(x[0]) + 0.01
    return x


SRC_FILES = ("problem.cpp", "problemS10.cpp", "problemG7.cpp")
WIND_NAMES = ("u", "v", "w", "du_dx", "du_dy", "du_dz", "dv_dx", "dv_dy", "dv_dz", "dw_dx", "dw_dy", "dw_dz")


def run_case(job):
    """One case, start to finish, in the reference's own calling order.  job: dict(tag, ci, mission, N, airframe, goal,
    start, kind) with kind "small" (non-shipped gains, perturbed air frames, table / grid wind) or "shipped"
    (everything as the repository ships it; wind models 1 and 0)."""
    t_start = time.time()
    it = cinterp.Interp([open(os.path.join(REF, "src", f), errors="replace").read() for f in SRC_FILES])
    ci, mission, N, airframe, goal, start, kind = (job[k] for k in ("ci", "mission", "N", "airframe", "goal", "start", "kind"))
    tag = job["tag"]
    out = {}
    rng = np.random.default_rng((4200 if kind == "small" else 7700) + ci)
    ac15 = np.array(read_param_values(os.path.join(REF, "aircraft", airframe + ".param")))
    lim8 = np.array(read_param_values(os.path.join(REF, "problems", mission, "limits.param")))
    assert len(ac15) == 15 and len(lim8) == 8
    if kind == "small":
        if ci >= 2:     # perturbed air-frame coefficients: mass, area, e, AR, Cd0 all away from the shipped values
            ac15[[0, 2, 3, 4, 5]] *= rng.uniform(0.8, 1.25, 5)
        # non-shipped gains, all different and non-zero: kT kp kv ka kdt
        gains = np.array([0.37, 5.3, 2.9, 0.0, 1.7]) * (1.0 + 0.1 * ci)
    else:
        gains = np.array(read_param_values(os.path.join(REF, "problems", mission, "gains.param")))
        assert len(gains) == 5
    o = RefProblem(it, mission, N, aircraft_members(ac15), gains, lim8, goal, start)      # runs setLimits, InitialCond, countG(x)
    x0 = np.array(o.x)
    neG = o.neG
    out[tag + "meta"] = np.array([0 if mission == "S10" else 1, N, o.n, o.neF, neG])
    out[tag + "ac15"], out[tag + "gains"], out[tag + "lim8"] = ac15, gains, lim8
    out[tag + "goal"], out[tag + "start"] = np.array(goal), np.array(start)
    out[tag + "chi_d"] = np.array([getattr(o, "chi_d", 0.0)])      # problemG7 declares it; problemS10 has none
    out[tag + "x0"] = x0
    out[tag + "xlow"], out[tag + "xupp"] = np.array(o.xlow), np.array(o.xupp)
    out[tag + "Flow"], out[tag + "Fupp"] = np.array(o.Flow), np.array(o.Fupp)
    out[tag + "iGfun"] = np.array(o.iGfun[:neG], dtype=np.int32)
    out[tag + "jGvar"] = np.array(o.jGvar[:neG], dtype=np.int32)
    if kind == "small":
        out[tag + "ioutput"] = np.frombuffer(it.output.files.get("Ioutput.txt", "").encode(), dtype=np.uint8)
    print("  %s %s N=%d %s: pattern after %.0f s" % (tag, mission, N, airframe, time.time() - t_start), flush=True)

    # test points: (wind kind, x, table)
    if kind == "small":
        grid_v = rng.uniform(-6, 6, (4, 4, 4))
        grid = dict(origin=(-260.0 + start[1], -240.0 + start[0], -30.0), spacing=(150.0, 150.0, 150.0), datum=(10.0, -20.0, 5.0))
        points = [("shear", x0.copy(), None), ("shear", perturbed(x0, N, rng), None)]
        for _ in range(3):
            tbl = rng.uniform(-0.3, 0.3, (12, N + 1))
            tbl[:3] *= 10.0
            points.append(("table", perturbed(x0, N, rng), tbl))
        points.append(("grid", perturbed(x0, N, rng), None))
        # round 3: (a) the linear boundary layer with Vref, href other than the 2.4 / 10 the reference hard-codes
        # (src/problem.cpp:504-505): handed to the reference as a model-99 table holding exactly what its case 1 would
        # compute with those two numbers (v = -Vref zs / href, dv_dz = -Vref / href, zs = -z_NED: :520-524); the
        # product's shear-wind kernels take (Vref, href) themselves.  (b) wind model 3 on a NON-cubic grid, 5 x 3 x 4.
        shear_pts = []
        for _ in range(3):
            xs = perturbed(x0, N, rng)
            Vref, href = float(rng.uniform(0.0, 5.0)), float(rng.uniform(5.0, 20.0))       # BASELINE configs[3]'s ranges
            tbl = np.zeros((12, N + 1))
            zs = -xs[1:].reshape(N + 1, 11)[:, 2]
            tbl[1] = -Vref * zs / href
            tbl[8] = -Vref / href
            points.append(("sheartable", xs, tbl))
            shear_pts.append((Vref, href))
        grid2_v = rng.uniform(-6, 6, (5, 3, 4))
        grid2 = dict(origin=(-260.0 + start[1], -240.0 + start[0], -30.0), spacing=(150.0, 150.0, 150.0), datum=(10.0, -20.0, 5.0))
        points.append(("grid2", perturbed(x0, N, rng), None))
    else:
        points = [("shear", x0.copy(), None), ("shear", perturbed(x0, N, rng), None), ("none", perturbed(x0, N, rng), None)]
    X, Fs, Gs, kinds, tables, wouts = [], [], [], [], [], []
    for pk, x, tbl in points:
        if pk == "shear":
            o.Pwindmodel = 1
        elif pk == "none":
            o.Pwindmodel = 0                        # src/problem.cpp:477-492
        elif pk in ("table", "sheartable"):
            o.set_wind_table(tbl)
        elif pk == "grid2":
            o.set_grid(grid2_v, grid2["origin"], grid2["spacing"], grid2["datum"])
        else:
            o.set_grid(grid_v, grid["origin"], grid["spacing"], grid["datum"])
        xl = [float(t) for t in x]
        F, G = [0.0] * o.neF, [0.0] * neG
        # src/DefineFG.cpp:24-37
        it.call(o, "modelWind", xl)
        it.call(o, "computeF", xl, F)
        it.call(o, "computeG", xl, G)
        X.append(x); Fs.append(F); Gs.append(G)
        kinds.append({"none": 0, "shear": 1, "table": 99, "grid": 3, "sheartable": 199, "grid2": 4}[pk])
        tables.append(np.array([getattr(o, nm) for nm in WIND_NAMES]))
        wouts.append(it.output.files.get("Woutput.txt", ""))
    out[tag + "X"], out[tag + "F"], out[tag + "G"] = np.array(X), np.array(Fs), np.array(Gs)
    out[tag + "windmodel"] = np.array(kinds)
    if kind == "small":
        out[tag + "wind"] = np.array(tables)          # the twelve member vectors as modelWind left them (ENU)
        out[tag + "grid_v"] = grid_v
        out[tag + "grid_geom"] = np.array(list(grid["origin"]) + list(grid["spacing"]) + list(grid["datum"]))
        out[tag + "woutput0"] = np.frombuffer(wouts[0].encode(), dtype=np.uint8)   # Woutput.txt of the first point
        out[tag + "shear"] = np.array(shear_pts)      # (Vref, href) of the windmodel-199 points, in order
        out[tag + "grid2_v"] = grid2_v
        out[tag + "grid2_geom"] = np.array(list(grid2["origin"]) + list(grid2["spacing"]) + list(grid2["datum"]))
    print("case %s %s N=%d %s: n=%d neF=%d neG=%d, %d points, %d interpreted calls, %.0f s" %
          (tag, mission, N, airframe, o.n, o.neF, neG, len(points), it.calls, time.time() - t_start), flush=True)
    return tag, out


def decode_pattern(iG, jG, neF, nb, pF=8, px=11):
    """(Fnum, xnum, tf, tx) that countG files for pattern entry (ii, jj): src/problem.cpp:826-849, and :883-910 for the
    dt slot of a dynamics row (re-targeted to tx = tf).  C integer division truncates toward zero.  Used by the "long"
    set only, where running countG's neF x n probe itself is out of reach; checked there against the arrays the
    interpreted countG leaves behind at a small ts."""
    def cdiv(a, b):
        return int(a / b)
    Fs, xs, tfs, txs = [], [], [], []
    for ii, jj in zip(iG, jG):
        ii, jj = int(ii), int(jj)
        Fnum = ii % pF
        if Fnum == 0 and ii != 0:
            Fnum = pF
        tf = cdiv(ii - 1, pF)
        if ii >= neF - nb:
            Fnum = pF + nb - (neF - 1 - ii)
        xnum = px if jj == 0 else (jj - 1) % px
        tx = cdiv(jj - 1, px)
        if jj == 0 and 1 <= Fnum <= pF:
            tx = tf
        Fs.append(float(Fnum)); xs.append(float(xnum)); tfs.append(float(tf)); txs.append(float(tx))
    return Fs, xs, tfs, txs


def closed_form_pattern(mission, N):
    """SURVEY.md section 8's closed form of the pattern, in row-major (row, column) order: the objective row, per node
    k and state r the 13 entries (8k+r, 0), (8k+r, 11k+1+m) m = 0..10, (8k+r, 11(k+1)+r), then the boundary rows.
    Asserted equal to what the interpreted countG finds at a small ts before it is used (run_long)."""
    iG, jG = [], []

    def add(i, cols):
        for j in cols:
            iG.append(i); jG.append(j)
    if mission == "S10":
        add(0, [0] + [c for k in range(N + 1) for c in (11 * k + 1, 11 * k + 2, 11 * k + 11)])
    else:
        add(0, [0, 1, 2] + [11 * k + 11 for k in range(N)] + [11 * N + 1, 11 * N + 2, 11 * N + 11])
    for k in range(N):
        for r in range(1, 9):
            add(8 * k + r, [0] + [11 * k + 1 + m for m in range(11)] + [11 * (k + 1) + r])
    base = 8 * N + 1
    if mission == "S10":
        for b in range(11):
            add(base + b, [0, 1 + b, 11 * N + 1 + b])
    else:
        for b in range(12):
            fnum = 9 + b
            if fnum in (9, 10, 20):
                add(base + b, [0, 1, 2, 11 * N + 1, 11 * N + 2])
            else:
                add(base + b, [0, fnum - 8, 11 * N + fnum - 8])
    return iG, jG


def run_long(job):
    """BASELINE configs[2] (S10, skywalker, ts = 2000).  countG's probe (352 million gradient calls) is out of reach, so
    the pattern comes from the closed form -- equal to countG's at every ts the other sets ran it for -- and the four
    sparse arrays from decode_pattern; InitialCond, setLimits, modelWind, computeF and computeG (214 037 gradient calls
    per point) are the reference's.  Stored: x, F, every 53rd G entry, and three sums over G."""
    t_start = time.time()
    it = cinterp.Interp([open(os.path.join(REF, "src", f), errors="replace").read() for f in SRC_FILES])
    mission, N, airframe, goal, start, tag = (job[k] for k in ("mission", "N", "airframe", "goal", "start", "tag"))
    nb = 11 if mission == "S10" else 12
    # the decode rule against an interpreted countG
    small = RefProblem(it, mission, 12, aircraft_members(np.array(read_param_values(os.path.join(REF, "aircraft", airframe + ".param")))),
                       np.array(read_param_values(os.path.join(REF, "problems", mission, "gains.param"))),
                       np.array(read_param_values(os.path.join(REF, "problems", mission, "limits.param"))), goal, start)
    iGs, jGs = closed_form_pattern(mission, 12)
    assert small.neG == len(iGs) and list(iGs) == small.iGfun[:small.neG] and list(jGs) == small.jGvar[:small.neG]
    dec = decode_pattern(iGs, jGs, small.neF, nb)
    for got, name in zip(dec, ("F_sparse", "x_sparse", "tf_sparse", "tx_sparse")):
        assert got == [float(v) for v in getattr(small, name)[:small.neG]], name
    print("  decode rule equals the interpreted countG's arrays at ts = 12 (%d entries), %.0f s" % (small.neG, time.time() - t_start), flush=True)

    rng = np.random.default_rng(9900)
    ac15 = np.array(read_param_values(os.path.join(REF, "aircraft", airframe + ".param")))
    lim8 = np.array(read_param_values(os.path.join(REF, "problems", mission, "limits.param")))
    gains = np.array(read_param_values(os.path.join(REF, "problems", mission, "gains.param")))
    iG, jG = closed_form_pattern(mission, N)
    neG = len(iG)
    # problem::problem runs (its neF x n work arrays, 352 million entries each, are held sparsely: cinterp.BigArray); of the
    # mission constructor's body, setLimits and InitialCond run, countG(x) is replaced by the closed form as explained
    o = RefProblem(it, mission, N, aircraft_members(ac15), gains, lim8, goal, start, derived_body=False)
    it.call(o, "setLimits")
    it.call(o, "InitialCond")
    x0 = np.array(o.x)
    o.neG = neG
    o.iGfun, o.jGvar = [int(v) for v in iG], [int(v) for v in jG]
    o.F_sparse, o.x_sparse, o.tf_sparse, o.tx_sparse = decode_pattern(iG, jG, o.neF, nb)
    out = {tag + "meta": np.array([0 if mission == "S10" else 1, N, o.n, o.neF, neG]), tag + "goal": np.array(goal),
           tag + "x0": x0, tag + "xlow": np.array(o.xlow), tag + "xupp": np.array(o.xupp),
           tag + "Flow": np.array(o.Flow), tag + "Fupp": np.array(o.Fupp)}
    X, Fs, Gsamp, Gsums = [], [], [], []
    weights = 1.0 + (np.arange(neG) % 1009) / 1009.0            # position-dependent: a permuted G changes this sum
    for x in (x0.copy(), perturbed(x0, N, rng)):
        xl = [float(t) for t in x]
        F, G = [0.0] * o.neF, [0.0] * neG
        it.call(o, "modelWind", xl)
        it.call(o, "computeF", xl, F)
        it.call(o, "computeG", xl, G)
        G = np.array(G)
        X.append(x); Fs.append(F); Gsamp.append(G[::53])
        Gsums.append([np.nansum(G), np.nansum(np.abs(G)), np.nansum(G * weights), float(np.isnan(G).sum())])
        print("  %s point done, %d interpreted calls, %.0f s" % (tag, it.calls, time.time() - t_start), flush=True)
    out[tag + "X"], out[tag + "F"] = np.array(X), np.array(Fs)
    out[tag + "G_every_53rd"], out[tag + "G_sums"] = np.array(Gsamp), np.array(Gsums)
    return tag, out


SMALL = [  # (mission, N, airframe, goal (east, north, up, radius), start)
    ("S10", 6, "tempest", (400.0, 0.0, 70.0, 100.0), (0.0, 0.0, 0.0)),
    ("G7", 6, "tempest", (400.0, 0.0, 70.0, 0.0), (0.0, 0.0, 0.0)),
    ("S10", 9, "skywalker", (250.0, -120.0, 70.0, 80.0), (15.0, -25.0, -40.0)),
    ("G7", 9, "skywalker", (300.0, 200.0, 70.0, 0.0), (-20.0, 30.0, -55.0)),
    # odd ts: c0 is odd, so the product takes its scalar-store kernels; 20 nodes: more than one 16-node store group
    ("S10", 13, "tempest", (380.0, 40.0, 70.0, 120.0), (5.0, 8.0, -35.0)),
    ("G7", 20, "tempest", (350.0, -150.0, 70.0, 0.0), (12.0, -7.0, -45.0)),
]
AIRFRAMES = ["tempest", "skywalker", "tempest_eric", "tempest_wences", "tempest_will"]
# the repository as shipped (ts = 100, shipped gains / limits / air frames), arguments `0 0 100 400 0 70 <radius> <airframe> <mission>`
# (src/arguments.cpp:36-44); plus BASELINE configs[1]'s ts = 200 once per mission
SHIPPED = [(m, 100, a, (400.0, 0.0, 70.0, 100.0 if m == "S10" else 0.0), (0.0, 0.0, 0.0)) for m in ("S10", "G7") for a in AIRFRAMES] + \
          [("S10", 200, "tempest", (400.0, 0.0, 70.0, 100.0), (0.0, 0.0, 0.0)), ("G7", 200, "tempest", (400.0, 0.0, 70.0, 0.0), (0.0, 0.0, 0.0))]


def json_keys():
    """The key names the reference's result writer assigns (problem::writeJSON, src/problem.cpp:1247-1365: statements of the form
    snopt_results["a"]["b"] = ...;), read from its text at fixture time.  Stored as names only: tests/golden/results_json_keys.json."""
    import json
    import re
    ref = os.environ.get("TOL_REFERENCE", "/root/reference")
    with open(os.path.join(ref, "src", "problem.cpp")) as fh:
        text = fh.read()
    start = text.index("problem::writeJSON")
    end = text.index("problem::writeTXT", start)
    first = text.count("\n", 0, start) + 1
    last = text.count("\n", 0, end) + 1
    keys = {"top": []}
    var = None
    for stmt in re.finditer(r'^\s*(\w+)((?:\s*\[\s*"[^"]+"\s*\])+)\s*=[^=]', text[start:end], flags=re.M):
        path = re.findall(r'"([^"]+)"', stmt.group(2))
        var = var or stmt.group(1)
        if stmt.group(1) != var:
            continue
        if path[0] not in keys["top"]:
            keys["top"].append(path[0])
        if len(path) == 2:
            keys.setdefault(path[0], [])
            if path[1] not in keys[path[0]]:
                keys[path[0]].append(path[1])
        elif len(path) > 2:
            raise SystemExit("deeper nesting than the extractor knows: %r" % (path,))
    out = {"source": "src/problem.cpp:%d-%d (problem::writeJSON), assignments to %s[...]" % (first, last, var), "keys": keys}
    path = os.environ.get("REF_VECTORS_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden", "results_json_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote", path, {k: len(v) for k, v in keys.items()})


def main():
    import argparse
    import multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", choices=["small", "shipped", "long", "json-keys"], default="small")
    ap.add_argument("--workers", type=int, default=6)
    args = ap.parse_args()
    t_start = time.time()
    if args.set == "json-keys":
        return json_keys()
    if args.set == "long":
        tag, out = run_long(dict(tag="l0_", mission="S10", N=2000, airframe="skywalker", goal=(400.0, 0.0, 70.0, 100.0), start=(0.0, 0.0, 0.0)))
        out["cases"] = np.array([tag])
        np.savez_compressed(OUT_LONG, **out)
        print("wrote", OUT_LONG, os.path.getsize(OUT_LONG), "bytes, %.0f s" % (time.time() - t_start))
        return
    spec = SMALL if args.set == "small" else SHIPPED
    prefix = "c" if args.set == "small" else "s"
    jobs = [dict(tag="%s%d_" % (prefix, ci), ci=ci, mission=m, N=N, airframe=a, goal=g, start=st, kind=args.set)
            for ci, (m, N, a, g, st) in enumerate(spec)]
    order = sorted(range(len(jobs)), key=lambda i: -jobs[i]["N"])          # longest first
    with mp.Pool(min(args.workers, len(jobs))) as pool:
        done = dict(pool.imap_unordered(run_case, [jobs[i] for i in order]))
    out = {}
    for j in jobs:
        out.update(done[j["tag"]])
    out["cases"] = np.array([j["tag"] for j in jobs])
    path = OUT if args.set == "small" else OUT_SHIPPED
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes, %.0f s" % (time.time() - t_start))


if __name__ == "__main__":
    main()
