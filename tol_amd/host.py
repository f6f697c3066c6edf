"""Python host layer over the C ABI of libtolfg.so (include/tolfg.h).

`Problem` is one SNOPT problem instance as the reference builds it in mission_select
(ref: src/tol.cpp:5-24): sizes, pattern, x0, bounds and the DEFINEGusrfg_ callback.
`Batch` is the device-resident evaluator for many independent trajectories (no reference
counterpart); its buffers are torch CUDA tensors, torch being used only for device memory and streams.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import Config, BatchConfig, Traj, check, lib

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _enc(s):
    return None if s is None else str(s).encode()


def _wind_grid(v, origin, spacing, datum):
    """tolfg_wind_grid from a [nx][ny][nz] array of the north wind component (keeps the array alive)."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    g = capi.WindGrid()
    g.nx, g.ny, g.nz = v.shape
    g.x0, g.y0, g.z0 = origin
    g.dx, g.dy, g.dz = spacing
    g.east_from_datum, g.north_from_datum, g.up_from_datum = datum
    g.v = _d(v)
    return g, v


class _DeviceBlock:
    """Owner of a tolfg_device_alloc block, seen by torch through __cuda_array_interface__; freed with the last tensor."""

    def __init__(self, ptr, shape, typestr, L):
        self.ptr = ptr
        self._L = L              # the library whose allocator made the block frees it
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}

    def __del__(self):
        try:
            if self.ptr:
                self._L.tolfg_device_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def _owned_tensor(ptr, shape, dtype, device, L):
    """A torch view of a tolfg_device_alloc block.  torch takes the device from the pointer's attributes; should it see the
    block anywhere else than on `device` it would hand back a COPY (and the placement would be lost without a word), so
    the pointer is checked and a mismatch is an error the caller can fall back from."""
    import torch
    blk = _DeviceBlock(ptr, shape, "<f8" if dtype == "f64" else "<f4", L)
    t = torch.as_tensor(blk, device=torch.device("cuda", device))
    if t.data_ptr() != ptr or t.device.index != device:
        del t
        raise capi.TolfgError(capi.ERR_HIP, "torch did not adopt the placed block in place (device attribution of the pointer)")
    return t


def device_alloc(shape, dtype="f64", device=0, library=None):
    """A device tensor [shape] in the library's placed form (tolfg_device_alloc); the tensor owns the memory."""
    L = library or lib()
    n = 1
    for d in shape:
        n *= int(d)
    ptr = C.c_void_p()
    check(L.tolfg_device_alloc(int(device), n * (8 if dtype == "f64" else 4), C.byref(ptr)), L)
    return _owned_tensor(ptr.value, tuple(int(d) for d in shape), dtype, device, L)


class Problem:
    """ref: `new problemS10(args)` / `new problemG7(args)` + what runSNOPT hands to SNOPT."""

    def __init__(self, mission, aircraft="tempest", east=0.0, north=0.0, up=100.0, east_goal=400.0,
                 north_goal=0.0, up_goal=70.0, radius_goal=100.0, ts=0, windmodel=capi.WIND_SHEAR,
                 Vref=2.4, href=10.0, start=(0.0, 0.0, 0.0), device=0, root_path=None, debug_dumps=False,
                 pattern="reference", persistent_arrays=False, library=None):
        L = self._L = library or lib()
        cfg = Config()
        L.tolfg_config_default(C.byref(cfg))
        self._keep = (_enc(mission), _enc(aircraft), _enc(root_path))
        cfg.mission, cfg.aircraft, cfg.root_path = self._keep
        cfg.east, cfg.north, cfg.up = east, north, up
        cfg.east_goal, cfg.north_goal, cfg.up_goal, cfg.radius_goal = east_goal, north_goal, up_goal, radius_goal
        cfg.ts, cfg.windmodel, cfg.Vref, cfg.href = int(ts), int(windmodel), Vref, href
        cfg.xi, cfg.yi, cfg.zi = start
        cfg.device, cfg.debug_dumps = int(device), int(bool(debug_dumps))
        cfg.pattern = capi.PATTERNS[pattern]
        cfg.persistent_arrays = int(bool(persistent_arrays))
        self._h = C.c_void_p()
        self._rc(L.tolfg_create(C.byref(cfg), C.byref(self._h)))
        n, neF, neG = C.c_int(), C.c_int(), C.c_int()
        self._rc(L.tolfg_sizes(self._h, C.byref(n), C.byref(neF), C.byref(neG)))
        self.n, self.neF, self.neG = n.value, neF.value, neG.value
        self.mission = mission

    def _rc(self, rc):
        return check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None):
            self._L.tolfg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- companion data for the SNOPT driver
    def pattern(self):
        iG = np.zeros(self.neG, dtype=np.int32)
        jG = np.zeros(self.neG, dtype=np.int32)
        self._rc(self._L.tolfg_pattern(self._h, _i(iG), _i(jG)))
        return iG, jG

    def x0(self):
        x = np.zeros(self.n)
        self._rc(self._L.tolfg_x0(self._h, _d(x)))
        return x

    def bounds(self):
        xl, xu = np.zeros(self.n), np.zeros(self.n)
        Fl, Fu = np.zeros(self.neF), np.zeros(self.neF)
        self._rc(self._L.tolfg_bounds(self._h, _d(xl), _d(xu), _d(Fl), _d(Fu)))
        return xl, xu, Fl, Fu

    def tolerances(self):
        a, b = C.c_double(), C.c_double()
        self._rc(self._L.tolfg_tolerances(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_wind_table(self, wind_enu):
        w = np.ascontiguousarray(wind_enu, dtype=np.float64)
        self._rc(self._L.tolfg_set_wind_table(self._h, _d(w)))

    def set_wind_grid(self, v, origin, spacing=(150.0, 150.0, 150.0), datum=(0.0, 0.0, 0.0)):
        """Wind model 3: gridded north wind component v[nx][ny][nz] on a regular ENU grid."""
        g, keep = _wind_grid(v, origin, spacing, datum)
        self._rc(self._L.tolfg_set_wind_grid(self._h, C.byref(g)))

    def write_json(self, filename, x, final_cost):
        """ref: problem::writeJSON -- the `snopt_results.json` the mission glue and MATLAB tools read."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._rc(self._L.tolfg_write_json(self._h, _d(x), float(final_cost), str(filename).encode()))

    # ---- the callback, exactly as SNOPT enters it
    def make_current(self):
        self._L.tolfg_set_current(self._h)

    def handle_index(self):
        return self._L.tolfg_handle_index(self._h)

    # ---- arrays used in place (include/tolfg.h): the caller's promise that x / F / G stay where they are
    def register_arrays(self, x=None, F=None, G=None):
        """Pin and map these numpy arrays (float64, contiguous, 16-byte aligned) for the kernel to use in place whenever
        define_fg(..., F=F, G=G) passes exactly them; call forget_arrays() before letting go of them."""
        ptr = lambda a: None if a is None else _d(a)      # noqa: E731
        self._rc(self._L.tolfg_register_arrays(self._h, ptr(x), ptr(F), ptr(G)))

    def forget_arrays(self):
        self._rc(self._L.tolfg_forget_arrays(self._h))

    def registered_arrays(self):
        k = self._L.tolfg_registered_arrays(self._h)
        if k < 0:
            self._rc(k)
        return k

    def define_fg(self, x, needF=True, needG=True, status=1, use_iu=False, F=None, G=None):
        """Call DEFINEGusrfg_ with snOptA's argument convention; returns (F, G, Status).  F and G are fresh arrays
        unless given (a caller that keeps its arrays, like SNOPT, passes the same ones every time)."""
        x = x if isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous else np.ascontiguousarray(x, dtype=np.float64)
        F = np.zeros(self.neF) if F is None else F
        G = np.zeros(self.neG) if G is None else G
        st, n, neF, neG = C.c_int(status), C.c_int(len(x)), C.c_int(self.neF), C.c_int(self.neG)
        nf, ng = C.c_int(int(needF)), C.c_int(int(needG))
        zero = C.c_int(0)
        if use_iu:
            iu = (C.c_int * 2)(capi.IU_MAGIC, self.handle_index())
            leniu = C.c_int(2)
            self._L.DEFINEGusrfg_(C.byref(st), C.byref(n), _d(x), C.byref(nf), C.byref(neF), _d(F), C.byref(ng),
                                C.byref(neG), _d(G), None, C.byref(zero), iu, C.byref(leniu), None, C.byref(zero))
        else:
            self.make_current()
            self._L.DEFINEGusrfg_(C.byref(st), C.byref(n), _d(x), C.byref(nf), C.byref(neF), _d(F), C.byref(ng),
                                C.byref(neG), _d(G), None, C.byref(zero), None, C.byref(zero), None, C.byref(zero))
        return F, G, st.value

    def time_callback(self, x, calls, warm=50, needF=True, needG=True, in_place=True):
        """Mean wall time (us) of one DEFINEGusrfg_ call entered from native code like snOptA enters it; F and G
        are the same arrays every call.  in_place: the arrays are registered for the duration (the contract a SNOPT driver
        enters with tolfg_register_arrays / persistent_arrays); False times the default contract, every call staged
        through the library's pinned buffers.  Returns (us_per_call, F, G)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        F, G = np.zeros(self.neF), np.zeros(self.neG)
        us = C.c_double()
        st = self._L.tolfg_time_callback_as(self._h, _d(x), _d(F), _d(G), int(needF), int(needG), int(bool(in_place)), int(warm), int(calls),
                                          C.byref(us))
        if st != 1:
            self._rc(st if st < 0 else capi.ERR_HIP)
        return us.value, F, G

    # ---- the three methods the reference's callback dispatches to
    def modelWind(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._rc(self._L.tolfg_modelWind(self._h, _d(x)))

    def computeF(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        F = np.zeros(self.neF)
        self._rc(self._L.tolfg_computeF(self._h, _d(x), _d(F)))
        return F

    def computeG(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        G = np.zeros(self.neG)
        self._rc(self._L.tolfg_computeG(self._h, _d(x), _d(G)))
        return G


@dataclass
class Trajectory:
    aircraft: int = 0
    mission: str = "S10"        # read by "mixed" batches only
    Vref: float = 2.4
    href: float = 10.0
    north_goal: float = 0.0
    east_goal: float = 400.0
    radius_goal: float = 100.0
    xi: float = 0.0
    yi: float = 0.0
    zi: float = 0.0


class Batch:
    """Device-resident evaluation of B trajectories that share ts (one GPU).  mission: "S10", "G7" or
    "mixed" -- then every Trajectory names its own mission, n/neF/neG are the row sizes to allocate for
    (the larger mission's) and sizes_of()/pattern(mission) give one mission's layout."""

    def __init__(self, mission, aircraft=("tempest",), ts=0, windmodel=capi.WIND_SHEAR, dtype="f64",
                 device=0, root_path=None, pattern="reference", library=None):
        L = self._L = library or lib()
        names = [a.encode() for a in aircraft]
        arr = (C.c_char_p * len(names))(*names)
        cfg = BatchConfig()
        self._keep = (_enc(mission), _enc(root_path), arr, names)
        cfg.mission, cfg.root_path = self._keep[0], self._keep[1]
        cfg.aircraft, cfg.n_aircraft = arr, len(names)
        cfg.ts, cfg.windmodel = int(ts), int(windmodel)
        cfg.dtype = capi.F64 if dtype == "f64" else capi.F32
        cfg.device = int(device)
        cfg.pattern = capi.PATTERNS[pattern]
        self._h = C.c_void_p()
        self._rc(L.tolfg_batch_create(C.byref(cfg), C.byref(self._h)))
        n, neF, neG = C.c_int(), C.c_int(), C.c_int()
        self._rc(L.tolfg_batch_sizes(self._h, C.byref(n), C.byref(neF), C.byref(neG)))
        self.n, self.neF, self.neG = n.value, neF.value, neG.value
        self.mission, self.dtype, self.device = mission, dtype, int(device)
        self.windmodel = windmodel
        self.B = 0
        self.missions = []

    def _rc(self, rc):
        return check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None):
            self._L.tolfg_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def N(self):
        return (self.n - 1) // 11 - 1

    def sizes_of(self, mission):
        """(n, neF, neG) of one mission of this batch."""
        n, neF, neG = C.c_int(), C.c_int(), C.c_int()
        self._rc(self._L.tolfg_batch_mission_sizes(self._h, capi.MISSIONS[mission], C.byref(n), C.byref(neF), C.byref(neG)))
        return n.value, neF.value, neG.value

    def pattern(self, mission=None):
        if mission is None:
            iG = np.zeros(self.neG, dtype=np.int32)
            jG = np.zeros(self.neG, dtype=np.int32)
            self._rc(self._L.tolfg_batch_pattern(self._h, _i(iG), _i(jG)))
            return iG, jG
        neG = self.sizes_of(mission)[2]
        iG = np.zeros(neG, dtype=np.int32)
        jG = np.zeros(neG, dtype=np.int32)
        self._rc(self._L.tolfg_batch_mission_pattern(self._h, capi.MISSIONS[mission], _i(iG), _i(jG)))
        return iG, jG

    def set_trajectories(self, trajs):
        arr = (Traj * len(trajs))()
        self.missions = [tr.mission if self.mission == "mixed" else self.mission for tr in trajs]
        for t, tr in enumerate(trajs):
            arr[t].aircraft, arr[t].Vref, arr[t].href = tr.aircraft, tr.Vref, tr.href
            arr[t].mission = capi.MISSIONS[tr.mission] if self.mission == "mixed" else 0
            arr[t].north_goal, arr[t].east_goal, arr[t].radius_goal = tr.north_goal, tr.east_goal, tr.radius_goal
            arr[t].xi, arr[t].yi, arr[t].zi = tr.xi, tr.yi, tr.zi
        self._rc(self._L.tolfg_batch_set_trajectories(self._h, len(trajs), arr))
        self.B = len(trajs)

    def set_wind_grid(self, v, origin, spacing=(150.0, 150.0, 150.0), datum=(0.0, 0.0, 0.0)):
        g, keep = _wind_grid(v, origin, spacing, datum)
        self._rc(self._L.tolfg_batch_set_wind_grid(self._h, C.byref(g)))
        self.windmodel = capi.WIND_GRID

    def x0(self, t, zi=0.0):
        x = np.zeros(self.n)
        self._rc(self._L.tolfg_batch_x0(self._h, int(t), float(zi), _d(x)))
        return x

    def bounds(self, t, zi=0.0):
        if not 0 <= int(t) < len(self.missions):
            raise capi.TolfgError(capi.ERR_ARG, f"trajectory {t} is not described (set_trajectories holds {len(self.missions)})")
        neF = self.sizes_of(self.missions[t])[1]
        xl, xu = np.zeros(self.n), np.zeros(self.n)
        Fl, Fu = np.zeros(neF), np.zeros(neF)
        self._rc(self._L.tolfg_batch_bounds(self._h, int(t), float(zi), _d(xl), _d(xu), _d(Fl), _d(Fu)))
        return xl, xu, Fl, Fu

    # ---- device buffers (torch is plumbing: memory + streams)
    def torch_dtype(self):
        import torch
        return torch.float64 if self.dtype == "f64" else torch.float32

    def alloc(self, B=None, pad=None, placed=None, tries=12):
        """X, F, G device tensors with row strides padded to `pad` elements (default: 16 bytes).
        placed (default: True when the trajectories are described and pad is the default): G comes from the library's
        placement-probing allocator (tolfg_batch_alloc_outputs: an address range backed by 2 MiB physical chunks, the best
        of up to `tries` candidates by the bare store loop of this batch's launch -- where G lands in HBM decides the class
        the launch runs in, include/tolfg.h); the probe times are kept in self.placement.  The tensor owns the memory."""
        import torch
        B = self.B if B is None else B
        v = pad if pad is not None else (2 if self.dtype == "f64" else 4)
        up = lambda m: (m + v - 1) // v * v   # noqa: E731
        dev = torch.device("cuda", self.device)
        dt = self.torch_dtype()
        X = torch.zeros((B, up(self.n)), dtype=dt, device=dev)
        F = torch.zeros((B, up(self.neF)), dtype=dt, device=dev)
        if placed is None:
            placed = pad is None and 0 < B <= getattr(self, "B", 0)
        G = None
        if placed:
            try:
                G = self.alloc_outputs(B, tries)
                G.zero_()
            except capi.TolfgError as exc:      # the buffer is a convenience, not a requirement: torch's allocator then
                G = None
                self.placement = {"allocator": "torch (the placed allocation was not available: %s)" % exc, "candidates": 0, "probe_us": []}
        if G is None:
            G = torch.zeros((B, up(self.neG)), dtype=dt, device=dev)
        return X, F, G

    def alloc_outputs(self, B, tries=12):
        """The G tensor [B][ldg] from tolfg_batch_alloc_outputs (uninitialised); self.placement records the probe."""
        import torch
        ptr, ldg, tried = C.c_void_p(), C.c_long(), C.c_int()
        probe = (C.c_double * max(int(tries), 1))()
        self._rc(self._L.tolfg_batch_alloc_outputs(self._h, int(B), int(tries), C.byref(ptr), C.byref(ldg), probe, C.byref(tried)))
        self.placement = {"allocator": "tolfg_batch_alloc_outputs: one address range backed by 2 MiB physical chunks; best of the "
                                       "candidates by the bare store loop of the launch's own shape",
                          "candidates": tried.value, "probe_us": [round(probe[i], 2) for i in range(tried.value)] if tried.value > 1 else []}
        return _owned_tensor(ptr.value, (int(B), ldg.value), self.dtype, self.device, self._L)

    def _check(self, t, name, rows, cols):
        """The C ABI takes raw pointers and strides: a tensor of the wrong dtype, device or layout would
        make the kernels read or write out of bounds, so it is refused here (TOLFG_ERR_ARG)."""
        import torch
        why = None
        if not isinstance(t, torch.Tensor):
            why = "is not a tensor"
        elif t.dtype != self.torch_dtype():
            why = f"has dtype {t.dtype}, the batch computes in {self.torch_dtype()}"
        elif t.device.type != "cuda" or t.device.index != self.device:
            why = f"lives on {t.device}, the batch on cuda:{self.device}"
        elif cols is None:
            if t.dim() != 1 or t.shape[0] < rows or t.stride(0) != 1:
                why = f"must be a contiguous vector of at least {rows} elements"
        elif t.dim() != 2 or t.shape[0] < rows or t.shape[1] < cols or t.stride(1) != 1 or (rows > 1 and t.stride(0) < cols):
            why = f"must be [>= {rows}][>= {cols}] with unit inner stride, got shape {tuple(t.shape)} strides {tuple(t.stride())}"
        if why:
            raise capi.TolfgError(capi.ERR_ARG, f"{name} {why}")

    def eval(self, X, F, G, wind=None, needF=True, needG=True, stream=None, B=None, obj=None):
        """Enqueue one evaluation on `stream` (default: torch's current stream).  `obj` (optional,
        B elements) also receives the objectives F[:, 0], contiguous."""
        import torch
        if not isinstance(X, torch.Tensor):
            raise capi.TolfgError(capi.ERR_ARG, "X is not a tensor")
        B = X.shape[0] if B is None else B
        self._check(X, "X", B, self.n)
        if needF:
            self._check(F, "F", B, self.neF)
        if needG:
            self._check(G, "G", B, self.neG)
        if wind is not None:
            if wind.dim() != 3 or not wind.is_contiguous() or tuple(wind.shape[1:]) != (12, self.N + 1):
                raise capi.TolfgError(capi.ERR_ARG, f"wind must be contiguous [B][12][{self.N + 1}], got {tuple(wind.shape)}")
            self._check(wind.view(wind.shape[0], -1), "wind", B, 12 * (self.N + 1))
        if obj is not None:
            self._check(obj, "obj", B, None)
        if stream is None:
            stream = torch.cuda.current_stream(X.device).cuda_stream
        Fp, Fs = (F.data_ptr(), F.stride(0)) if needF else (None, 0)
        Gp, Gs = (G.data_ptr(), G.stride(0)) if needG else (None, 0)
        self._rc(self._L.tolfg_batch_eval(self._h, int(B), X.data_ptr(), X.stride(0), Fp, Fs,
                                     Gp, Gs, None if wind is None else wind.data_ptr(),
                                     int(needF), int(needG), None if obj is None else obj.data_ptr(),
                                     C.c_void_p(stream)))

    def x0_device(self, X, stream=None, B=None):
        """Initial guess of trajectories [0,B) written into the rows of the device tensor X."""
        import torch
        B = X.shape[0] if B is None else B
        self._check(X, "X", B, self.n)
        if stream is None:
            stream = torch.cuda.current_stream(X.device).cuda_stream
        self._rc(self._L.tolfg_batch_x0_device(self._h, int(B), X.data_ptr(), X.stride(0), C.c_void_p(stream)))

    def bounds_device(self, xlow, xupp, Flow, Fupp, stream=None, B=None):
        import torch
        B = xlow.shape[0] if B is None else B
        for t, name, cols in ((xlow, "xlow", self.n), (xupp, "xupp", self.n), (Flow, "Flow", self.neF), (Fupp, "Fupp", self.neF)):
            self._check(t, name, B, cols)
        if xlow.stride(0) != xupp.stride(0) or Flow.stride(0) != Fupp.stride(0):
            raise capi.TolfgError(capi.ERR_ARG, "xlow/xupp and Flow/Fupp must share their row strides")
        if stream is None:
            stream = torch.cuda.current_stream(xlow.device).cuda_stream
        self._rc(self._L.tolfg_batch_bounds_device(self._h, int(B), xlow.data_ptr(), xupp.data_ptr(), xlow.stride(0),
                                              Flow.data_ptr(), Fupp.data_ptr(), Flow.stride(0), C.c_void_p(stream)))

    def objectives(self, F, out=None, stream=None, B=None):
        import torch
        B = F.shape[0] if B is None else B
        self._check(F, "F", B, 1)
        if out is None:
            out = torch.empty(B, dtype=F.dtype, device=F.device)
        self._check(out, "out", B, None)
        if stream is None:
            stream = torch.cuda.current_stream(F.device).cuda_stream
        self._rc(self._L.tolfg_batch_objectives(self._h, int(B), F.data_ptr(), F.stride(0), out.data_ptr(),
                                           C.c_void_p(stream)))
        return out

    def status(self):
        """Raises TolfgError(ERR_HIP) when an evaluation since the last call lost an objective partial (ask after the
        evaluations have completed, e.g. after torch.cuda.synchronize())."""
        self._rc(self._L.tolfg_batch_status(self._h))

    def set_timing(self, on=True):
        self._rc(self._L.tolfg_batch_set_timing(self._h, int(bool(on))))

    def set_store_shape(self, on=True):
        """Measurement aid: while on, eval() launches the bare store loop of its own launch shape (F, G: garbage)."""
        self._rc(self._L.tolfg_batch_set_store_shape(self._h, int(bool(on))))

    def kernel_time(self):
        """(launches, avg_ms, min_ms) of fg_kernel since the last call (HIP events on the launch stream)."""
        a, m = C.c_double(), C.c_double()
        n = self._L.tolfg_batch_kernel_time(self._h, C.byref(a), C.byref(m))
        if n < 0:
            self._rc(n)
        return n, a.value, m.value

    def algorithmic_bytes(self, B=None):
        return self._L.tolfg_batch_algorithmic_bytes(self._h, int(self.B if B is None else B))


class Multi:
    """One process, several GPUs (include/tolfg.h section 4): a batch sharded over `devices`, one launch per device,
    the objectives gathered with ncclAllGather.  The device buffers belong to the library; fetch() copies a shard's F and
    G to the host (tests, small batches)."""

    def __init__(self, mission, aircraft=("tempest",), ts=0, windmodel=capi.WIND_SHEAR, dtype="f64", devices=(0,),
                 root_path=None, pattern="reference", library=None):
        L = self._L = library or lib()
        names = [a.encode() for a in aircraft]
        arr = (C.c_char_p * len(names))(*names)
        cfg = BatchConfig()
        self._keep = (_enc(mission), _enc(root_path), arr, names)
        cfg.mission, cfg.root_path = self._keep[0], self._keep[1]
        cfg.aircraft, cfg.n_aircraft = arr, len(names)
        cfg.ts, cfg.windmodel = int(ts), int(windmodel)
        cfg.dtype = capi.F64 if dtype == "f64" else capi.F32
        cfg.device = int(devices[0])
        cfg.pattern = capi.PATTERNS[pattern]
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        self._rc(L.tolfg_multi_create(C.byref(cfg), devs, len(devices), C.byref(self._h)))
        n, neF, neG = C.c_int(), C.c_int(), C.c_int()
        self._rc(L.tolfg_multi_sizes(self._h, C.byref(n), C.byref(neF), C.byref(neG)))
        self.n, self.neF, self.neG = n.value, neF.value, neG.value
        self.mission, self.dtype, self.devices = mission, dtype, tuple(int(d) for d in devices)
        self.total = 0

    def _rc(self, rc):
        return check(rc, self._L)

    def close(self):
        if getattr(self, "_h", None):
            self._L.tolfg_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rccl_library(self):
        return self._L.tolfg_multi_rccl_library().decode()

    def set_trajectories(self, trajs):
        arr = (Traj * len(trajs))()
        for t, tr in enumerate(trajs):
            arr[t].aircraft, arr[t].Vref, arr[t].href = tr.aircraft, tr.Vref, tr.href
            arr[t].mission = capi.MISSIONS[tr.mission] if self.mission == "mixed" else 0
            arr[t].north_goal, arr[t].east_goal, arr[t].radius_goal = tr.north_goal, tr.east_goal, tr.radius_goal
            arr[t].xi, arr[t].yi, arr[t].zi = tr.xi, tr.yi, tr.zi
        self._rc(self._L.tolfg_multi_set_trajectories(self._h, len(trajs), arr))
        self.total = len(trajs)

    def shard(self, i):
        lo, hi = C.c_long(), C.c_long()
        self._rc(self._L.tolfg_multi_shard(self._h, int(i), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def set_wind_grid(self, v, origin, spacing=(150.0, 150.0, 150.0), datum=(0.0, 0.0, 0.0)):
        g, keep = _wind_grid(v, origin, spacing, datum)
        self._rc(self._L.tolfg_multi_set_wind_grid(self._h, C.byref(g)))

    def set_wind_tables(self, wind_enu):
        """wind_enu: [total][12][ts+1], the table of every trajectory in global order (batch created with WIND_TABLE)."""
        w = np.ascontiguousarray(wind_enu, dtype=np.float64)
        N = (self.n - 1) // 11 - 1
        if w.shape != (self.total, 12, N + 1):
            raise capi.TolfgError(capi.ERR_ARG, f"wind tables must be [{self.total}][12][{N + 1}], got {w.shape}")
        self._rc(self._L.tolfg_multi_set_wind_tables(self._h, _d(w)))

    def x0(self):
        self._rc(self._L.tolfg_multi_x0(self._h))

    def eval(self, needF=True, needG=True):
        self._rc(self._L.tolfg_multi_eval(self._h, int(needF), int(needG)))

    def gather_objectives(self):
        out = np.zeros(self.total, dtype=np.float64 if self.dtype == "f64" else np.float32)
        self._rc(self._L.tolfg_multi_gather_objectives(self._h, out.ctypes.data))
        return out

    # ---- the asynchronous form: evaluations run beside the gathers of the ones before them
    def _xptrs(self, dX):
        if dX is None:
            return None, 0
        if len(dX) != len(self.devices):
            raise capi.TolfgError(capi.ERR_ARG, "one X pointer per device is required")
        return (C.c_void_p * len(dX))(*[int(p) for p in dX]), len(dX)

    def eval_from(self, dX, needF=True, needG=True):
        """One evaluation, device i reading its shard's x rows from the device pointer dX[i] (row stride as buffers(i))."""
        arr, n = self._xptrs(dX)
        self._rc(self._L.tolfg_multi_eval_from(self._h, arr, n, int(needF), int(needG)))

    def gather_begin(self):
        t = C.c_ulong()
        self._rc(self._L.tolfg_multi_gather_begin(self._h, C.byref(t)))
        return t.value

    def gather_wait(self, ticket, want=True):
        out = np.zeros(self.total, dtype=np.float64 if self.dtype == "f64" else np.float32) if want else None
        self._rc(self._L.tolfg_multi_gather_wait(self._h, int(ticket), None if out is None else out.ctypes.data))
        return out

    def step(self, dX=None, needG=True):
        """eval (F always) + gather_begin; returns the gather's ticket."""
        arr, n = self._xptrs(dX)
        t = C.c_ulong()
        self._rc(self._L.tolfg_multi_step(self._h, arr, n, int(needG), C.byref(t)))
        return t.value

    def set_issue(self, mode):
        """'grouped' (one ncclGroupStart/End bracket from the caller's thread) | 'threads' (every device's thread calls for
        its own communicator)."""
        self._rc(self._L.tolfg_multi_set_issue(self._h, {"grouped": capi.ISSUE_GROUPED, "threads": capi.ISSUE_THREADS}[mode]))

    def set_gather(self, mode):
        """'rccl' (ncclAllGather into a device vector on every device) | 'host' (the finalizing waves store the objectives straight
        into one pinned host vector: no collective in the step)."""
        self._rc(self._L.tolfg_multi_set_gather(self._h, {"rccl": capi.GATHER_RCCL, "host": capi.GATHER_HOST}[mode]))

    def set_placement(self, tries):
        self._rc(self._L.tolfg_multi_set_placement(self._h, int(tries)))

    def buffers(self, i):
        """((dX, ldx), (dF, ldf), (dG, ldg)) of shard i: device pointers (ints) on devices[i] and row strides in elements."""
        dX, dF, dG = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ldx, ldf, ldg = C.c_long(), C.c_long(), C.c_long()
        self._rc(self._L.tolfg_multi_buffers(self._h, int(i), C.byref(dX), C.byref(ldx), C.byref(dF), C.byref(ldf), C.byref(dG), C.byref(ldg)))
        return (dX.value, ldx.value), (dF.value, ldf.value), (dG.value, ldg.value)

    def time_steps(self, steps, warm=10, x_sets=None, needF=True, needG=True, gather=True):
        """The native step loop (tolfg_multi_time_steps).  x_sets: a list of per-device pointer lists used in rotation, or
        None for the library's own X.  Returns a dict of the timings (us) and the per-device launch times."""
        nd = len(self.devices)
        n_x = 0 if not x_sets else len(x_sets)
        arr = None
        if n_x:
            flat = [int(p) for xs in x_sets for p in xs]
            if len(flat) != n_x * nd:
                raise capi.TolfgError(capi.ERR_ARG, "every X set needs one pointer per device")
            arr = (C.c_void_p * len(flat))(*flat)
        t = capi.MultiTiming()
        per = (C.c_double * nd)()
        self._rc(self._L.tolfg_multi_time_steps(self._h, n_x, arr, int(needF), int(needG), int(gather), int(warm), int(steps), C.byref(t), per))
        return {"wall_us_per_step": t.wall_us_per_step, "launch_us_per_step": t.launch_us_per_step, "issue_us_per_step": t.issue_us_per_step,
                "gather_us": t.gather_us, "devices": t.devices, "steps": t.steps, "issue": ("grouped", "threads")[t.issue],
                "gather": ("rccl", "host")[t.gather],
                "launch_us_per_device": [per[i] for i in range(nd)]}

    def rccl_version(self):
        return self._L.tolfg_multi_rccl_version()

    def mean_objective(self):
        m = C.c_double()
        self._rc(self._L.tolfg_multi_mean_objective(self._h, C.byref(m)))
        return m.value

    def sync(self):
        self._rc(self._L.tolfg_multi_sync(self._h))

    def fetch(self, i, with_x=False):
        """(F, G) -- with_x: (X, F, G) -- of shard i as host arrays [rows][ld] (device-to-host copies with the HIP runtime
        the library uses)."""
        dX, dF, dG = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ldx, ldf, ldg = C.c_long(), C.c_long(), C.c_long()
        self._rc(self._L.tolfg_multi_buffers(self._h, int(i), C.byref(dX), C.byref(ldx), C.byref(dF), C.byref(ldf), C.byref(dG), C.byref(ldg)))
        self.sync()
        lo, hi = self.shard(i)
        dt = np.float64 if self.dtype == "f64" else np.float32
        out = []
        hip = capi._hip_runtime
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipSetDevice(self.devices[i])
        for ptr, ld in ((dX, ldx.value),) * bool(with_x) + ((dF, ldf.value), (dG, ldg.value)):
            a = np.zeros((hi - lo, ld), dtype=dt)
            rc = hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2)       # hipMemcpyDeviceToHost
            if rc != 0:
                raise capi.TolfgError(capi.ERR_HIP, f"hipMemcpy failed with {rc}")
            out.append(a)
        return tuple(out)
