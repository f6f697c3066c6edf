"""ctypes declarations for include/tolfg.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

OK, ERR_ARG, ERR_PARAM, ERR_HIP, ERR_NOCURRENT = 0, -1, -2, -3, -4
WIND_NONE, WIND_SHEAR, WIND_GRID, WIND_TABLE = 0, 1, 3, 99
F64, F32 = 0, 1
PATTERN_REFERENCE, PATTERN_COMPACT = 0, 1
PATTERNS = {"reference": 0, "compact": 1}
MISSION_S10, MISSION_G7 = 0, 1
MISSIONS = {"S10": 0, "G7": 1}
IU_MAGIC = 0x70F6

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class TolfgError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"tolfg error {code}: {text}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("mission", C.c_char_p), ("aircraft", C.c_char_p), ("root_path", C.c_char_p),
                ("east", C.c_double), ("north", C.c_double), ("up", C.c_double),
                ("east_goal", C.c_double), ("north_goal", C.c_double), ("up_goal", C.c_double),
                ("radius_goal", C.c_double),
                ("ts", C.c_int), ("windmodel", C.c_int),
                ("Vref", C.c_double), ("href", C.c_double),
                ("xi", C.c_double), ("yi", C.c_double), ("zi", C.c_double),
                ("device", C.c_int), ("debug_dumps", C.c_int), ("pattern", C.c_int), ("persistent_arrays", C.c_int)]


class Traj(C.Structure):
    _fields_ = [("aircraft", C.c_int), ("mission", C.c_int),
                ("Vref", C.c_double), ("href", C.c_double),
                ("north_goal", C.c_double), ("east_goal", C.c_double), ("radius_goal", C.c_double),
                ("xi", C.c_double), ("yi", C.c_double), ("zi", C.c_double)]


class WindGrid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("x0", C.c_double), ("y0", C.c_double), ("z0", C.c_double),
                ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("east_from_datum", C.c_double), ("north_from_datum", C.c_double), ("up_from_datum", C.c_double),
                ("v", _dp)]


class BatchConfig(C.Structure):
    _fields_ = [("mission", C.c_char_p), ("root_path", C.c_char_p),
                ("aircraft", C.POINTER(C.c_char_p)), ("n_aircraft", C.c_int),
                ("ts", C.c_int), ("windmodel", C.c_int), ("dtype", C.c_int), ("device", C.c_int),
                ("pattern", C.c_int)]


class MultiTiming(C.Structure):
    _fields_ = [("wall_us_per_step", C.c_double), ("launch_us_per_step", C.c_double), ("issue_us_per_step", C.c_double),
                ("gather_us", C.c_double), ("devices", C.c_int), ("steps", C.c_int), ("issue", C.c_int), ("gather", C.c_int)]


MULTI_SLOTS = 4
ISSUE_GROUPED, ISSUE_THREADS = 0, 1
GATHER_RCCL, GATHER_HOST = 0, 1
_vpp = C.POINTER(C.c_void_p)

# snFunA, include/snopt/snopt.h:60-66 of the reference
SNFUNA = C.CFUNCTYPE(None, _ip, _ip, _dp, _ip, _ip, _dp, _ip, _ip, _dp, C.c_char_p, _ip, _ip, _ip, _dp, _ip)

# every symbol include/tolfg.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tolfg_last_error": (C.c_char_p, []),
    "tolfg_version": (C.c_char_p, []),
    "tolfg_default_root": (C.c_char_p, []),
    "tolfg_read_params": (C.c_int, [C.c_char_p, _dp, C.c_int]),
    "tolfg_config_default": (None, [C.POINTER(Config)]),
    "tolfg_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "tolfg_destroy": (None, [C.c_void_p]),
    "tolfg_sizes": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    "tolfg_pattern": (C.c_int, [C.c_void_p, _ip, _ip]),
    "tolfg_x0": (C.c_int, [C.c_void_p, _dp]),
    "tolfg_bounds": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "tolfg_tolerances": (C.c_int, [C.c_void_p, _dp, _dp]),
    "tolfg_set_wind_table": (C.c_int, [C.c_void_p, _dp]),
    "tolfg_set_wind_grid": (C.c_int, [C.c_void_p, C.POINTER(WindGrid)]),
    "tolfg_batch_set_wind_grid": (C.c_int, [C.c_void_p, C.POINTER(WindGrid)]),
    "tolfg_write_json": (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_char_p]),
    "tolfg_set_current": (None, [C.c_void_p]),
    "tolfg_get_current": (C.c_void_p, []),
    "tolfg_handle_index": (C.c_int, [C.c_void_p]),
    "DEFINEGusrfg_": (None, [_ip, _ip, _dp, _ip, _ip, _dp, _ip, _ip, _dp, C.c_char_p, _ip, _ip, _ip, _dp, _ip]),
    "tolfg_register_arrays": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "tolfg_forget_arrays": (C.c_int, [C.c_void_p]),
    "tolfg_registered_arrays": (C.c_int, [C.c_void_p]),
    "tolfg_time_callback": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "tolfg_time_callback_as": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "tolfg_modelWind": (C.c_int, [C.c_void_p, _dp]),
    "tolfg_computeF": (C.c_int, [C.c_void_p, _dp, _dp]),
    "tolfg_computeG": (C.c_int, [C.c_void_p, _dp, _dp]),
    "tolfg_batch_create": (C.c_int, [C.POINTER(BatchConfig), C.POINTER(C.c_void_p)]),
    "tolfg_batch_destroy": (None, [C.c_void_p]),
    "tolfg_batch_sizes": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    "tolfg_batch_pattern": (C.c_int, [C.c_void_p, _ip, _ip]),
    "tolfg_batch_mission_sizes": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip]),
    "tolfg_batch_mission_pattern": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip]),
    "tolfg_batch_set_trajectories": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Traj)]),
    "tolfg_batch_x0": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _dp]),
    "tolfg_batch_x0_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p]),
    "tolfg_batch_bounds_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p,
                                            C.c_long, C.c_void_p]),
    "tolfg_batch_bounds": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _dp, _dp, _dp, _dp]),
    "tolfg_batch_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                   C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tolfg_batch_status": (C.c_int, [C.c_void_p]),
    "tolfg_batch_objectives": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]),
    "tolfg_batch_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "tolfg_batch_kernel_time": (C.c_int, [C.c_void_p, _dp, _dp]),
    "tolfg_batch_set_store_shape": (C.c_int, [C.c_void_p, C.c_int]),
    "tolfg_device_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tolfg_device_free": (C.c_int, [C.c_void_p]),
    "tolfg_batch_alloc_outputs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_long), _dp, _ip]),
    "tolfg_batch_algorithmic_bytes": (C.c_double, [C.c_void_p, C.c_int]),
    "tolfg_multi_create": (C.c_int, [C.POINTER(BatchConfig), _ip, C.c_int, C.POINTER(C.c_void_p)]),
    "tolfg_multi_destroy": (None, [C.c_void_p]),
    "tolfg_multi_sizes": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    "tolfg_multi_set_trajectories": (C.c_int, [C.c_void_p, C.c_long, C.POINTER(Traj)]),
    "tolfg_multi_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "tolfg_multi_buffers": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_long), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_long), C.POINTER(C.c_void_p), C.POINTER(C.c_long)]),
    "tolfg_multi_set_wind_grid": (C.c_int, [C.c_void_p, C.POINTER(WindGrid)]),
    "tolfg_multi_set_wind_tables": (C.c_int, [C.c_void_p, _dp]),
    "tolfg_multi_x0": (C.c_int, [C.c_void_p]),
    "tolfg_multi_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "tolfg_multi_gather_objectives": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tolfg_multi_eval_from": (C.c_int, [C.c_void_p, _vpp, C.c_int, C.c_int, C.c_int]),
    "tolfg_multi_gather_begin": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulong)]),
    "tolfg_multi_gather_wait": (C.c_int, [C.c_void_p, C.c_ulong, C.c_void_p]),
    "tolfg_multi_step": (C.c_int, [C.c_void_p, _vpp, C.c_int, C.c_int, C.POINTER(C.c_ulong)]),
    "tolfg_multi_set_issue": (C.c_int, [C.c_void_p, C.c_int]),
    "tolfg_multi_set_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "tolfg_multi_set_placement": (C.c_int, [C.c_void_p, C.c_int]),
    "tolfg_multi_time_steps": (C.c_int, [C.c_void_p, C.c_int, _vpp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(MultiTiming), _dp]),
    "tolfg_multi_rccl_version": (C.c_int, []),
    "tolfg_measurement_build": (C.c_int, []),
    "tolfg_multi_mean_objective": (C.c_int, [C.c_void_p, _dp]),
    "tolfg_multi_sync": (C.c_int, [C.c_void_p]),
    "tolfg_multi_rccl_library": (C.c_char_p, []),
    "tolfg_shard_bounds": (C.c_int, [C.c_long, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "tolfg_compact_gathered": (C.c_int, [C.c_void_p, C.c_size_t, C.c_long, C.c_int, C.c_void_p]),
}

_lib = None
_measure_lib = None
_hip_runtime = None


def lib_path():
    """The product library; TOLFG_LIBRARY names another build of it (same-box A/B of two builds, tools/; the GPU suite
    against the measurement build)."""
    return os.environ.get("TOLFG_LIBRARY") or os.path.join(HERE, "lib", "libtolfg.so")


def measure_lib_path():
    """The measurement build: the same sources with the measurement variables of csrc/knobs.h compiled in."""
    return os.path.join(HERE, "lib", "libtolfg_measure.so")


def hip_runtime_path():
    """The single HIP runtime this process uses.

    A process may hold only ONE copy of libamdhip64: a second copy cannot open the GPU.  PyTorch
    wheels bundle their own copy, so when torch is importable its copy is the one (torch is imported
    first so that it is already mapped); otherwise the system runtime under /opt/rocm is used.
    TOLFG_HIP_RUNTIME overrides both."""
    env = os.environ.get("TOLFG_HIP_RUNTIME")
    if env:
        return env
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return cand
    except ImportError:
        pass
    for cand in ("/opt/rocm/lib/libamdhip64.so", "/opt/rocm/lib/libamdhip64.so.7"):
        if os.path.exists(cand):
            return cand
    raise TolfgError(ERR_HIP, "no libamdhip64 found (set TOLFG_HIP_RUNTIME)")


def mapped_hip_runtimes():
    """Paths of every libamdhip64 currently mapped into this process (should be exactly one)."""
    out = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    out.add(line.split()[-1])
    except OSError:
        pass
    return sorted(out)


def _load(path):
    """Load one build of the library.  Fails loudly when it has not been built (no fallback path exists)."""
    if not os.path.exists(path):
        raise TolfgError(ERR_HIP, f"{path} is missing: run `python -m tol_amd.build` (hipcc, gfx950)")
    # libtolfg.so carries no DT_NEEDED for the HIP runtime (csrc/Makefile explains why): make the
    # process-wide copy globally visible first, then load the library against it.
    global _hip_runtime
    if _hip_runtime is None:
        _hip_runtime = C.CDLL(hip_runtime_path(), mode=C.RTLD_GLOBAL)
    L = C.CDLL(path)
    if len(mapped_hip_runtimes()) > 1:
        raise TolfgError(ERR_HIP, "two HIP runtimes are mapped: %s" % mapped_hip_runtimes())
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib():
    """The product library (tol_amd/lib/libtolfg.so, or the build TOLFG_LIBRARY names)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if os.environ.get("TOLFG_LIBRARY"):      # a redirection of the whole library is worth one line
            import sys
            sys.stderr.write(f"tol_amd: TOLFG_LIBRARY is set: loading {path} instead of the shipped library\n")
        _lib = _load(path)
    return _lib


def measure_lib():
    """The measurement build, for tools and A/B tests: pass it as `library=` to Problem / Batch / Multi.  Objects remember
    the library that made them, so both builds can serve one process side by side."""
    global _measure_lib
    if _measure_lib is None:
        _measure_lib = _load(measure_lib_path())
    return _measure_lib


def check(rc, L=None):
    if rc != OK:
        raise TolfgError(rc, (L or lib()).tolfg_last_error().decode())
    return rc
