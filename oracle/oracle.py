"""ctypes window onto oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (tol_amd/) never does.  See oracle/tolfg_oracle.h for the parity status of the oracle.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "tol_amd", "data")

S10, G7 = 0, 1
WIND_NONE, WIND_SHEAR, WIND_GRID, WIND_TABLE = 0, 1, 3, 99
MISSION_ID = {"S10": S10, "G7": G7}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OrcProblem(C.Structure):
    _fields_ = [("mission", C.c_int), ("N", C.c_int),
                ("mm", C.c_double), ("SS", C.c_double), ("Cd0", C.c_double), ("AR", C.c_double),
                ("ee", C.c_double),
                ("kT", C.c_double), ("kp", C.c_double), ("kv", C.c_double), ("kdt", C.c_double),
                ("xg", C.c_double), ("yg", C.c_double), ("rg", C.c_double),
                ("chi_d", C.c_double),
                ("windmodel", C.c_int), ("Vref", C.c_double), ("href", C.c_double),
                ("wind", _dp),
                ("gnx", C.c_int), ("gny", C.c_int), ("gnz", C.c_int),
                ("gx0", C.c_double), ("gy0", C.c_double), ("gz0", C.c_double),
                ("gdx", C.c_double), ("gdy", C.c_double), ("gdz", C.c_double),
                ("gE", C.c_double), ("gN", C.c_double), ("gU", C.c_double),
                ("gv", _dp)]


def build(opt="O2"):
    """Compile the oracle with its Makefile (gcc).  Building the checker is not using it."""
    target = "liboracle.so" if opt == "O2" else "liboracle_O0.so"
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)
    return os.path.join(HERE, target)


_libs = {}


def lib(opt="O2"):
    if opt in _libs:
        return _libs[opt]
    path = os.path.join(HERE, "liboracle.so" if opt == "O2" else "liboracle_O0.so")
    if not os.path.exists(path):
        build(opt)
    L = C.CDLL(path)
    L.orc_read_params.argtypes = [C.c_char_p, _dp, C.c_int]
    L.orc_read_params.restype = C.c_int
    for name in ("orc_nb", "orc_n"):
        getattr(L, name).argtypes = [C.c_int]
        getattr(L, name).restype = C.c_int
    for name in ("orc_neF", "orc_neG", "orc_c0"):
        getattr(L, name).argtypes = [C.c_int, C.c_int]
        getattr(L, name).restype = C.c_int
    L.orc_pattern_closed.argtypes = [C.c_int, C.c_int, _ip, _ip]
    L.orc_pattern_closed.restype = None
    L.orc_pattern_walk.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip, _ip, _ip, _ip]
    L.orc_pattern_walk.restype = C.c_int
    L.orc_dispatch_closed.argtypes = [C.c_int, C.c_int, _ip, _ip, _ip, _ip]
    L.orc_dispatch_closed.restype = None
    L.orc_x0.argtypes = [C.POINTER(OrcProblem), C.c_double, C.c_double, C.c_double, _dp]
    L.orc_x0.restype = None
    L.orc_bounds.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double, C.c_double,
                             _dp, _dp, _dp, _dp]
    L.orc_bounds.restype = None
    L.orc_eval.argtypes = [C.POINTER(OrcProblem), _dp, C.c_int, _dp, C.c_int, _dp]
    L.orc_eval.restype = None
    L.orc_eval_entrywise.argtypes = [C.POINTER(OrcProblem), _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                     _ip, _ip, _ip, _ip]
    L.orc_eval_entrywise.restype = None
    L.orc_eval_batch.argtypes = [C.POINTER(OrcProblem), C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                 C.c_int]
    L.orc_eval_batch.restype = C.c_int
    _libs[opt] = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def read_params(path, maxn=64):
    out = np.zeros(maxn)
    cnt = lib().orc_read_params(path.encode(), _d(out), maxn)
    if cnt < 0:
        raise FileNotFoundError(path)
    return out[:min(cnt, maxn)].copy(), cnt


def sizes(mission, N):
    m = MISSION_ID[mission] if isinstance(mission, str) else mission
    L = lib()
    return L.orc_n(N), L.orc_neF(m, N), L.orc_neG(m, N)


class Problem:
    """One trajectory problem as the oracle sees it (mission, airframe, goal, wind)."""

    def __init__(self, mission, aircraft="tempest", N=None, east_goal=400.0, north_goal=0.0,
                 radius_goal=100.0, start=(0.0, 0.0, 0.0), windmodel=WIND_SHEAR, Vref=2.4, href=10.0,
                 wind_table=None, data_root=DATA, gains=None, wind_grid=None, ac15=None, lim8=None):
        self.mission = mission
        self.mid = MISSION_ID[mission]
        self.aircraft = aircraft
        self.ac15, cnt = read_params(os.path.join(data_root, "aircraft", aircraft + ".param"))
        if cnt != 15:
            raise ValueError("aircraft file must hold 15 values")   # src/parameters.cpp:45-67
        g, cnt = read_params(os.path.join(data_root, "problems", mission, "gains.param"))
        if cnt != 5:
            raise ValueError("gains file must hold 5 values")
        self.lim8, cnt = read_params(os.path.join(data_root, "problems", mission, "limits.param"))
        if cnt != 8:
            raise ValueError("limits file must hold 8 values")
        sn, cnt = read_params(os.path.join(data_root, "problems", mission, "snopt.param"))
        if cnt != 6:
            raise ValueError("snopt file must hold 6 values")
        if gains is not None:
            g = np.asarray(gains, dtype=float)
        if ac15 is not None:      # air-frame / limit values given directly (fixtures with non-shipped coefficients)
            self.ac15 = np.asarray(ac15, dtype=float).copy()
            assert self.ac15.shape == (15,)
        if lim8 is not None:
            self.lim8 = np.asarray(lim8, dtype=float).copy()
            assert self.lim8.shape == (8,)
        self.gains = g
        self.N = int(sn[0]) if N is None else int(N)
        self.opt_tol, self.feas_tol = sn[4], sn[5]
        self.start = tuple(float(v) for v in start)
        self.n, self.neF, self.neG = sizes(mission, self.N)
        self.nb = lib().orc_nb(self.mid)
        self.c0 = lib().orc_c0(self.mid, self.N)
        p = OrcProblem()
        p.mission, p.N = self.mid, self.N
        p.mm, p.SS, p.ee, p.AR, p.Cd0 = self.ac15[0], self.ac15[2], self.ac15[3], self.ac15[4], self.ac15[5]
        p.kT, p.kp, p.kv, p.kdt = g[0], g[1], g[2], g[4]
        # goals ENU -> NED, src/problem.cpp:24-27
        p.xg, p.yg, p.rg = north_goal, east_goal, radius_goal
        p.chi_d = math.atan2(p.yg - self.start[1], p.xg - self.start[0])   # src/problemG7.cpp:524
        p.windmodel, p.Vref, p.href = windmodel, Vref, href
        self._wind = None
        if wind_table is not None:
            self._wind = np.ascontiguousarray(wind_table, dtype=np.float64)
            assert self._wind.shape == (12, self.N + 1)
            p.wind = _d(self._wind)
            p.windmodel = WIND_TABLE
        if wind_grid is not None:
            # dict: v[nx][ny][nz], origin (x0,y0,z0), spacing (dx,dy,dz), datum (E,N,U)  -- wind model 3
            self._gv = np.ascontiguousarray(wind_grid["v"], dtype=np.float64)
            p.gnx, p.gny, p.gnz = self._gv.shape
            p.gx0, p.gy0, p.gz0 = wind_grid["origin"]
            p.gdx, p.gdy, p.gdz = wind_grid["spacing"]
            p.gE, p.gN, p.gU = wind_grid["datum"]
            p.gv = _d(self._gv)
            p.windmodel = WIND_GRID
        self.c = p

    # ---- setup pieces
    def x0(self):
        x = np.zeros(self.n)
        lib().orc_x0(C.byref(self.c), *self.start, _d(x))
        return x

    def bounds(self):
        xl, xu = np.zeros(self.n), np.zeros(self.n)
        Fl, Fu = np.zeros(self.neF), np.zeros(self.neF)
        lib().orc_bounds(self.mid, self.N, _d(self.ac15), _d(self.lim8), *self.start,
                         _d(xl), _d(xu), _d(Fl), _d(Fu))
        return xl, xu, Fl, Fu

    def pattern(self, walk=False):
        iG = np.zeros(self.neG, dtype=np.int32)
        jG = np.zeros(self.neG, dtype=np.int32)
        if walk:
            cap = self.neG + 64
            iG = np.zeros(cap, dtype=np.int32)
            jG = np.zeros(cap, dtype=np.int32)
            cnt = lib().orc_pattern_walk(self.mid, self.N, _i(iG), _i(jG), None, None, None, None)
            return iG[:cnt].copy(), jG[:cnt].copy()
        lib().orc_pattern_closed(self.mid, self.N, _i(iG), _i(jG))
        return iG, jG

    def dispatch(self, walk=False):
        arrs = [np.zeros(self.neG + 64, dtype=np.int32) for _ in range(4)]
        if walk:
            cnt = lib().orc_pattern_walk(self.mid, self.N, None, None, *[_i(a) for a in arrs])
            return [a[:cnt].copy() for a in arrs]
        lib().orc_dispatch_closed(self.mid, self.N, *[_i(a) for a in arrs])
        return [a[:self.neG].copy() for a in arrs]

    # ---- evaluation
    def eval(self, x, needF=True, needG=True, opt="O2"):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.n,)
        F, G = np.zeros(self.neF), np.zeros(self.neG)
        lib(opt).orc_eval(C.byref(self.c), _d(x), int(needF), _d(F), int(needG), _d(G))
        return F, G

    def eval_entrywise(self, x, needF=True, needG=True, opt="O2", dispatch=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        F, G = np.zeros(self.neF), np.zeros(self.neG)
        d = dispatch if dispatch is not None else self.dispatch()
        lib(opt).orc_eval_entrywise(C.byref(self.c), _d(x), int(needF), _d(F), int(needG), _d(G),
                                    self.neG, *[_i(a) for a in d])
        return F, G

    def undefined_mask(self):
        """True where the reference leaves G undefined: S10 boundary rows x dt column
        (src/problemS10.cpp:397)."""
        m = np.zeros(self.neG, dtype=bool)
        if self.mission == "S10":
            base = self.c0 + 104 * self.N
            m[base + 3 * np.arange(11)] = True
        return m


# tabG indices (0..10 = node variables, 11 = dt) each dynamics row of the reference assigns
# (src/problem.cpp:1080-1186); everything else in a row stays at its initial 0.0 for every x.
ASSIGNED_TABG = {1: (0, 3, 4, 5, 11), 2: (1, 3, 4, 5, 11), 3: (2, 3, 4, 11), 4: (3, 4, 5, 7, 10, 11),
                 5: (3, 4, 5, 6, 7, 11), 6: (3, 4, 5, 6, 7, 11), 7: (6, 8, 11), 8: (7, 9, 11)}


def compact_index(problem):
    """Indices into the reference-pattern G of the entries that are not structurally zero: per node
    the 46 entries the reference's dynamicsGradients can make non-zero (38 assigned + 8 next-node
    ones), the whole objective row, and the boundary rows without their dt column."""
    iG, jG = problem.pattern()
    N = problem.N
    keep = np.ones(problem.neG, dtype=bool)
    for e in range(problem.neG):
        row, col = int(iG[e]), int(jG[e])
        if 1 <= row <= 8 * N:
            k, r = (row - 1) // 8, (row - 1) % 8 + 1
            if col == 0:
                m = 11
            elif col >= 11 * (k + 1) + 1:
                continue                      # next-node entry, always 1.0
            else:
                m = col - (11 * k + 1)
            keep[e] = m in ASSIGNED_TABG[r]
        elif row > 8 * N:
            keep[e] = col != 0
    return np.flatnonzero(keep)


def eval_batch(problems, X, nthreads=1, opt="O2"):
    """Evaluate a list of oracle Problems (same N; missions may differ: rows are then sized for the larger
    mission and each row holds its own mission's neF / neG entries) on the rows of X."""
    B = len(problems)
    p0 = problems[0]
    X = np.ascontiguousarray(X, dtype=np.float64)
    assert X.shape == (B, p0.n) and all(p.n == p0.n for p in problems)
    arr = (OrcProblem * B)(*[p.c for p in problems])
    neF, neG = max(p.neF for p in problems), max(p.neG for p in problems)
    F = np.zeros((B, neF))
    G = np.zeros((B, neG))
    used = lib(opt).orc_eval_batch(arr, B, _d(X), p0.n, _d(F), neF, _d(G), neG, int(nthreads))
    return F, G, used


def perturbed(problem, seed, scale=0.05):
    """Seeded test point: SURVEY.md section 8(c) recipe -- every variable moved by
    scale*U(-1,1)*(1+|x|), then z in [-70,-30], Va in [12,18], T in [5,15] so that wind and drag
    terms are exercised."""
    rng = np.random.default_rng(seed)
    x = problem.x0()
    x = x + scale * rng.uniform(-1, 1, x.shape) * (1 + np.abs(x))
    node = x[1:].reshape(problem.N + 1, 11)
    node[:, 2] = rng.uniform(-70, -30, problem.N + 1)
    node[:, 3] = rng.uniform(12, 18, problem.N + 1)
    node[:, 10] = rng.uniform(5, 15, problem.N + 1)
    x[0] = abs(x[0]) + 0.01
    return x
