"""ctypes binding of libnbody_amd.so (include/nbody_amd.h + include/nbody_amd_ext.h).  No compute happens in Python and there is no
fallback: a missing library raises at import of the symbol table, a missing GPU raises NBodyError(NB_ERR_NO_DEVICE)."""
import ctypes as C
import os

import numpy as np

NB_OK = 0
NB_ERR_INVALID, NB_ERR_NO_DEVICE, NB_ERR_HIP, NB_ERR_STATE, NB_ERR_NOMEM, NB_ERR_IO = -1, -2, -3, -4, -5, -6
NB_F64, NB_F32, NB_F32_ACC64 = 0, 1, 2
NB_SCN_MIN_DIST, NB_SCN_FIRST_HIT, NB_SCN_MISSILE = 0, 1, 2
NB_MAX_WATCH = 16

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class NbConfig(C.Structure):
    _fields_ = [("n", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32), ("f64_large_min", C.c_int32),
                ("f64_split", C.c_int32), ("flags", C.c_int32), ("G", C.c_double), ("eps", C.c_double),
                ("dt", C.c_double)]


class NbScenario(C.Structure):
    _fields_ = [("kind", C.c_int32), ("first_step", C.c_int32), ("last_step", C.c_int32), ("planet", C.c_int32),
                ("asteroid", C.c_int32), ("n_watch", C.c_int32), ("watch", C.c_int32 * NB_MAX_WATCH),
                ("sync_every", C.c_int32), ("engine", C.c_int32), ("flags", C.c_int32), ("graph_chunk", C.c_int32),
                ("planet_radius", C.c_double), ("missile_speed", C.c_double)]


class NbScenarioResult(C.Structure):
    _fields_ = [("min_dist2", C.c_double), ("hit_step", C.c_int32), ("steps_done", C.c_int32),
                ("arrival_step", C.c_int32 * NB_MAX_WATCH), ("missile_cost", C.c_double * NB_MAX_WATCH)]


class NbStateHeader(C.Structure):
    _fields_ = [("n", C.c_int64), ("precision", C.c_int32), ("step", C.c_int32), ("planet", C.c_int32),
                ("asteroid", C.c_int32), ("G", C.c_double), ("eps", C.c_double), ("dt", C.c_double)]


class NbAnswer(C.Structure):
    _fields_ = [("min_dist", C.c_double), ("hit_time_step", C.c_int32), ("gravity_device_id", C.c_int32),
                ("missile_cost", C.c_double)]


class NbSolveOptions(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("engine", C.c_int32), ("streams", C.c_int32), ("p3_parallel", C.c_int32),
                ("graph_chunk", C.c_int32), ("handoff", C.c_int32), ("reserved", C.c_int32 * 2)]


class NbShardedRank(C.Structure):
    _fields_ = [("device", C.c_int32), ("compute_units", C.c_int32), ("first_target", C.c_int64), ("targets", C.c_int64),
                ("exchange", C.c_int32), ("comm_ranks", C.c_int32), ("comm_rank", C.c_int32), ("comm_device", C.c_int32),
                ("pci_bus_id", C.c_char * 16), ("uuid", C.c_char * 36), ("name", C.c_char * 64)]


class NbLaunchF32(C.Structure):
    _fields_ = [("src", C.c_void_p), ("out", C.c_void_p), ("vel", C.c_void_p), ("pos64", C.c_void_p),
                ("vel64", C.c_void_p), ("acc", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("n_src", C.c_int64), ("tgt_off", C.c_int64),
                ("n_tgt", C.c_int64), ("eps2", C.c_float), ("dt", C.c_float), ("acc64", C.c_int32),
                ("targets_per_lane", C.c_int32), ("j_split", C.c_int32), ("source_path", C.c_int32),
                ("wg_size", C.c_int32), ("phase", C.c_int32), ("src_begin", C.c_int64), ("src_end", C.c_int64),
                ("tgt", C.c_void_p)]


# every symbol include/nbody_amd.h (the run_step boundary) and include/nbody_amd_ext.h (raw launches, shared pairs, nb_sharded_*,
# nb_solve_ex) declare: (restype, argtypes)
SYMBOLS = {
    "nb_abi_version": (C.c_int, []),
    "nb_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nb_config_default": (C.c_int, [C.POINTER(NbConfig)]),
    "nb_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(NbConfig)]),
    "nb_destroy": (C.c_int, [C.c_void_p]),
    "nb_strerror": (C.c_char_p, [C.c_int]),
    "nb_last_error": (C.c_char_p, [C.c_void_p]),
    "nb_set_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p]),
    "nb_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]),
    "nb_set_mass": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "nb_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "nb_run_step": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p]),
    "nb_accel": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "nb_step_timed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "nb_run_scenario": (C.c_int, [C.c_void_p, C.POINTER(NbScenario), C.POINTER(NbScenarioResult)]),
    "nb_run_scenarios_batched": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(NbScenario), C.POINTER(NbScenarioResult),
                                          C.c_int]),
    "nb_restore_snapshot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "nb_save_state": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "nb_load_state": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "nb_state_file_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nb_read_state_file": (C.c_int, [C.c_char_p, C.POINTER(NbStateHeader), C.c_int64, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                    _u8p]),
    "nb_write_state_file": (C.c_int, [C.c_char_p, C.POINTER(NbStateHeader), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p]),
    "nb_solve": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p,
                           C.POINTER(C.c_int), C.c_int, C.POINTER(NbAnswer)]),
    "nb_solve_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p,
                              C.POINTER(C.c_int), C.c_int, C.POINTER(NbSolveOptions), C.POINTER(NbAnswer)]),
    "nb_launch_step_f32": (C.c_int, [C.POINTER(NbLaunchF32), C.c_void_p]),
    "nb_launch_accel_f32": (C.c_int, [C.POINTER(NbLaunchF32), C.c_void_p]),
    "nb_kernel_name_f32": (C.c_char_p, [C.POINTER(NbLaunchF32), C.c_int]),
    "nb_plan_f32": (C.c_int, [C.POINTER(NbLaunchF32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nb_workspace_bytes_f32": (C.c_int64, [C.c_int64, C.c_int]),
    "nb_workspace_bytes_sym_f32": (C.c_int64, [C.c_int64, C.c_int]),
    "nb_launch_pair_forces_f32": (C.c_int, [C.POINTER(NbLaunchF32), C.c_void_p]),
    "nb_launch_kick_drift_f32": (C.c_int, [C.POINTER(NbLaunchF32), C.c_int, C.c_void_p]),
    "nb_workspace_bytes_shared_pairs_f32": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "nb_plan_shared_pairs_f32": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nb_context_kernel_name": (C.c_char_p, [C.c_void_p]),
    "nb_selftest_pair_schedule": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "nb_selftest_pair_schedule_within": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_char_p, C.c_int]),
    "nb_sharded_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int64, C.c_int, C.c_double,
                                   C.c_double, C.c_double, C.c_int]),
    "nb_sharded_destroy": (C.c_int, [C.c_void_p]),
    "nb_sharded_set_deadline": (C.c_int, [C.c_void_p, C.c_double]),
    "nb_sharded_last_error": (C.c_char_p, [C.c_void_p]),
    "nb_sharded_set_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "nb_sharded_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp]),
    "nb_sharded_step": (C.c_int, [C.c_void_p, C.c_int]),
    "nb_sharded_save_state": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "nb_sharded_load_state": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "nb_sharded_step_timed": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "nb_sharded_step_profiled": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "nb_sharded_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nb_sharded_rank_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(NbShardedRank)]),
    "nb_sharded_kernel_name": (C.c_char_p, [C.c_void_p]),
}
# include/nbody_amd_debug.h: exported by the instrumented build (libnbody_amd_stamps.so) only
DEBUG_SYMBOLS = {
    "nb_enable_step_stamps": (C.c_int, [C.c_void_p, C.c_int]),
    "nb_read_step_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "nb_create_cu_masked": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(NbConfig), C.c_int]),
}
NB_EXCHANGE_RCCL, NB_EXCHANGE_COPY, NB_EXCHANGE_HOST = 1, 2, 3
NB_SHARDED_OVERLAP = 1
NB_CU_ALL, NB_CU_LOW, NB_CU_HIGH, NB_CU_EVEN, NB_CU_ODD = 0, 1, 2, 3, 4
NB_HANDOFF_AUTO, NB_HANDOFF_HOST_STAGED = 0, 1
NB_SHARDED_COPY_EXCHANGE = 2
NB_SHARDED_ORDERED_PAIRS = 4
NB_SHARDED_HOST_EXCHANGE = 8
NB_CFG_ORDERED_PAIRS = 1


class NBodyError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = f"{where}: {_strerror(code)} ({code})"
        if detail:
            msg += f" — {detail}"
        super().__init__(msg)


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnbody_amd.so")


def stamps_library_path():
    """The instrumented build (make stamps): the per-step fp64 kernel records clock stamps (nb_enable_step_stamps)."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnbody_amd_stamps.so")


_lib = None


def _share_hip_runtime_with_torch():
    """One process must not initialise two HIP runtimes.  libnbody_amd.so needs `libamdhip64.so.7`; a PyTorch-ROCm wheel
    ships its own copy and loads it by path.  If this library pulled in /opt/rocm's copy first, a later `import torch`
    would bring up a second runtime that reports "No HIP GPUs are available".  So when torch is installed but not yet
    imported, load ITS runtime first (cheap: no torch import); our library then binds to it by SONAME and a later
    `import torch` finds it already loaded.  Pure C/C++ hosts (bin/hw5, bin/nbody_bench) use /opt/rocm's runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load(path):
    if not os.path.exists(path):
        raise ImportError(f"{path} not built — run `make` (there is no CPU fallback)")
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(L, name)  # AttributeError if the ABI lost a symbol
        f.restype, f.argtypes = res, args
    for name, (res, args) in DEBUG_SYMBOLS.items():  # the instrumented build's extras, when this is that build
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
    return L


def _debug_symbol(name, ctx=None):
    """A symbol of include/nbody_amd_debug.h; the product library does not export it."""
    L = lib()
    if not hasattr(L, name):
        raise NBodyError(NB_ERR_STATE, name, "this build of the library carries no measurement hooks: use "
                         "libnbody_amd_stamps.so (make stamps; capi.use_library(capi.stamps_library_path()))")
    return getattr(L, name)


def lib():
    """Load libnbody_amd.so.  Raises if it has not been built (`make` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        _lib = _load(library_path())
    return _lib


class use_library:
    """`with capi.use_library(path):` — everything inside goes through another build of the library (measurement tools:
    the instrumented stamps build, same-device A/B of two kernel builds).  Contexts must not outlive the block."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        self.saved, _lib = _lib, _load(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved


def _strerror(code):
    try:
        return lib().nb_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "error"


def _check(rc, where, ctx=None):
    if rc != NB_OK:
        detail = lib().nb_last_error(ctx).decode()  # ctx None: the calling thread's last context-free failure
        raise NBodyError(rc, where, detail)


def device_count():
    n = C.c_int(0)
    lib().nb_device_count(C.byref(n))
    return n.value


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class Context:
    """One nb_context: a system of n bodies resident on one GPU."""

    def __init__(self, n, precision=NB_F64, device=0, G=None, eps=None, dt=None, f64_large_min=0, f64_split=0,
                 cu_mask=NB_CU_ALL, ordered_pairs=False, workspace_gib=0):
        cfg = NbConfig()
        _check(lib().nb_config_default(C.byref(cfg)), "nb_config_default")
        cfg.n, cfg.precision, cfg.device = n, precision, device
        cfg.f64_large_min, cfg.f64_split = f64_large_min, f64_split
        cfg.flags = (NB_CFG_ORDERED_PAIRS if ordered_pairs else 0) | ((int(workspace_gib) & 0xffff) << 8)  # NB_CFG_WORKSPACE_GIB
        if G is not None:
            cfg.G = G
        if eps is not None:
            cfg.eps = eps
        if dt is not None:
            cfg.dt = dt
        self.cfg, self.n = cfg, n
        self._h = C.c_void_p()
        if cu_mask != NB_CU_ALL:  # measurement knob of the instrumented build (include/nbody_amd_debug.h)
            rc = _debug_symbol("nb_create_cu_masked")(C.byref(self._h), C.byref(cfg), cu_mask)
        else:
            rc = lib().nb_create(C.byref(self._h), C.byref(cfg))
        if rc != NB_OK:
            h, self._h = self._h, C.c_void_p()
            detail = lib().nb_last_error(h).decode() if h else ""
            if h:
                lib().nb_destroy(h)
            raise NBodyError(rc, "nb_create", detail)

    def close(self):
        if self._h:
            lib().nb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_state(self, q, v, m, is_device=None):
        """q, v: (3, n) arrays (rows = the reference's qx,qy,qz / vx,vy,vz vectors); m: (n,)."""
        keep = [_d(q[0]), _d(q[1]), _d(q[2]), _d(v[0]), _d(v[1]), _d(v[2]), _d(m)]
        dev = None
        if is_device is not None:
            dev_arr = np.ascontiguousarray(is_device, dtype=np.uint8)
            dev = dev_arr.ctypes.data_as(_u8p)
        _check(lib().nb_set_state(self._h, *[p for _, p in keep], dev), "nb_set_state", self._h)

    def get_state(self):
        q = np.empty((3, self.n))
        v = np.empty((3, self.n))
        ptrs = [q[k].ctypes.data_as(_dp) for k in range(3)] + [v[k].ctypes.data_as(_dp) for k in range(3)]
        _check(lib().nb_get_state(self._h, *ptrs), "nb_get_state", self._h)
        return q, v

    def set_mass(self, index, m):
        _check(lib().nb_set_mass(self._h, index, m), "nb_set_mass", self._h)

    def step(self, first_step, count=1):
        _check(lib().nb_step(self._h, first_step, count), "nb_step", self._h)

    def run_step(self, step, qx, qy, qz, vx, vy, vz, m, is_device=None):
        """nb_run_step: the reference's run_step(step, n, qx, ..., m, type) as one call — the six state vectors (contiguous
        float64 arrays) are updated IN PLACE."""
        for a in (qx, qy, qz, vx, vy, vz):
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == (self.n,)):
                raise ValueError("run_step updates its arrays in place: contiguous float64 arrays of n elements")
        keep_m = _d(m)
        dev = None
        if is_device is not None:
            dev_arr = np.ascontiguousarray(is_device, dtype=np.uint8)
            dev = dev_arr.ctypes.data_as(_u8p)
        ptrs = [a.ctypes.data_as(_dp) for a in (qx, qy, qz, vx, vy, vz)]
        _check(lib().nb_run_step(self._h, step, *ptrs, keep_m[1], dev), "nb_run_step", self._h)

    def step_timed(self, first_step, count):
        ms = C.c_float()
        _check(lib().nb_step_timed(self._h, first_step, count, C.byref(ms)), "nb_step_timed", self._h)
        return ms.value

    def accel(self, step):
        a = np.empty((3, self.n))
        _check(lib().nb_accel(self._h, step, *[a[k].ctypes.data_as(_dp) for k in range(3)]), "nb_accel", self._h)
        return a

    def kernel_name(self):
        """Force kernel nb_step / nb_accel of this fp32 context launch (asks for K1s' workspace like the first step would)."""
        return lib().nb_context_kernel_name(self._h).decode()

    def last_error(self):
        """Text of the context's last failure — or note: a context that had to fall back from K1s to K1 says so here."""
        return lib().nb_last_error(self._h).decode()

    def enable_step_stamps(self, slots):
        _check(_debug_symbol("nb_enable_step_stamps")(self._h, slots), "nb_enable_step_stamps", self._h)

    def read_step_stamps(self, slots):
        """-> (slots, 2) uint64: GPU wall clock (100 MHz ticks) at entry / after the last store of each step launch."""
        out = np.zeros((slots, 2), dtype=np.uint64)
        _check(_debug_symbol("nb_read_step_stamps")(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), slots),
               "nb_read_step_stamps", self._h)
        return out

    def run_scenario(self, kind, planet, asteroid, first_step=0, last_step=200000, watch=(), sync_every=2000,
                     planet_radius=1e7, missile_speed=1e6, engine=0, flags=0, graph_chunk=0):
        s = _scenario_struct(kind, planet, asteroid, first_step, last_step, watch, sync_every, planet_radius,
                             missile_speed, engine, flags, graph_chunk)
        r = NbScenarioResult()
        _check(lib().nb_run_scenario(self._h, C.byref(s), C.byref(r)), "nb_run_scenario", self._h)
        return dict(min_dist2=r.min_dist2, hit_step=r.hit_step, steps_done=r.steps_done,
                    arrival_step=list(r.arrival_step[:len(watch)]), missile_cost=list(r.missile_cost[:len(watch)]))

    def save_state(self, path, step=0):
        _check(lib().nb_save_state(self._h, os.fsencode(path), step), "nb_save_state", self._h)

    def load_state(self, path):
        step = C.c_int()
        _check(lib().nb_load_state(self._h, os.fsencode(path), C.byref(step)), "nb_load_state", self._h)
        return step.value

    def restore_snapshot_from(self, src, slot):
        _check(lib().nb_restore_snapshot(self._h, src._h, slot), "nb_restore_snapshot", self._h)


NB_SCN_NO_SNAPSHOT = 1
NB_SCN_EAGER = 2


def _scenario_struct(kind, planet, asteroid, first_step=0, last_step=200000, watch=(), sync_every=2000,
                     planet_radius=1e7, missile_speed=1e6, engine=0, flags=0, graph_chunk=0):
    s = NbScenario()
    s.kind, s.first_step, s.last_step, s.planet, s.asteroid = kind, first_step, last_step, planet, asteroid
    s.n_watch = len(watch)
    for k, w in enumerate(watch):
        s.watch[k] = w
    s.sync_every, s.planet_radius, s.missile_speed, s.engine = sync_every, planet_radius, missile_speed, engine
    s.flags, s.graph_chunk = flags, graph_chunk
    return s


def run_scenarios_batched(contexts, scenarios):
    """contexts: list of Context (same n, same GPU); scenarios: list of dicts with run_scenario's keyword arguments.
    One launch per step serves all of them.  -> list of result dicts."""
    n = len(contexts)
    hs = (C.c_void_p * n)(*[c._h for c in contexts])
    ss = (NbScenario * n)(*[_scenario_struct(**kw) for kw in scenarios])
    rs = (NbScenarioResult * n)()
    _check(lib().nb_run_scenarios_batched(hs, ss, rs, n), "nb_run_scenarios_batched", contexts[0]._h)
    return [dict(min_dist2=r.min_dist2, hit_step=r.hit_step, steps_done=r.steps_done,
                 arrival_step=list(r.arrival_step[:len(kw.get("watch", ()))]),
                 missile_cost=list(r.missile_cost[:len(kw.get("watch", ()))])) for r, kw in zip(rs, scenarios)]


def state_file_info(path):
    """-> (n, precision, step) of a binary state file written by nb_save_state."""
    n, prec, step = C.c_int64(), C.c_int(), C.c_int()
    _check(lib().nb_state_file_info(os.fsencode(path), C.byref(n), C.byref(prec), C.byref(step)), "nb_state_file_info")
    return n.value, prec.value, step.value


def read_state_file(path):
    """-> (header dict, q (3,n), v (3,n), m (n,), is_device (n,) uint8) of a binary state file; no GPU involved."""
    h = NbStateHeader()
    _check(lib().nb_read_state_file(os.fsencode(path), C.byref(h), 0, None, None, None, None, None, None, None, None),
           "nb_read_state_file")
    n = h.n
    q, v, m, dev = np.empty((3, n)), np.empty((3, n)), np.empty(n), np.empty(n, dtype=np.uint8)
    ptrs = [q[k].ctypes.data_as(_dp) for k in range(3)] + [v[k].ctypes.data_as(_dp) for k in range(3)]
    _check(lib().nb_read_state_file(os.fsencode(path), C.byref(h), n, *ptrs, m.ctypes.data_as(_dp),
                                    dev.ctypes.data_as(_u8p)), "nb_read_state_file")
    hdr = dict(n=n, precision=h.precision, step=h.step, planet=h.planet, asteroid=h.asteroid, G=h.G, eps=h.eps, dt=h.dt)
    return hdr, q, v, m, dev


def write_state_file(path, q, v, m, is_device=None, planet=-1, asteroid=-1, precision=NB_F64, step=0, G=6.674e-11,
                     eps=1e-3, dt=60.0):
    """Binary (NBODYST2) form of the reference's text input (nbody.cc:22-39): bin/hw5 accepts it as <input>."""
    n = len(m)
    h = NbStateHeader(n, precision, step, planet, asteroid, G, eps, dt)
    keep = [_d(q[0]), _d(q[1]), _d(q[2]), _d(v[0]), _d(v[1]), _d(v[2]), _d(m)]
    dev = None
    if is_device is not None:
        dev_arr = np.ascontiguousarray(is_device, dtype=np.uint8)
        dev = dev_arr.ctypes.data_as(_u8p)
    _check(lib().nb_write_state_file(os.fsencode(path), C.byref(h), *[p for _, p in keep], dev), "nb_write_state_file")


_ENGINES = {None: 0, "auto": 0, "steps": 1, "persistent": 2}
_STREAMS = {None: 0, "auto": 0, "merged": 1, "split": 2}


def solve(n, planet, asteroid, q, v, m, is_device, devices=None, engine=None, streams=None, max_batch=0,
          p3_parallel=0, graph_chunk=0, handoff=NB_HANDOFF_AUTO):
    """The whole reference program (P1, P2, P3) on the GPU: nb_solve_ex.  The keyword arguments after `devices` are
    nb_solve_options (engine: steps|persistent, streams: merged|split; 0/None = the library's defaults)."""
    keep = [_d(q[0]), _d(q[1]), _d(q[2]), _d(v[0]), _d(v[1]), _d(v[2]), _d(m)]
    dev_arr = np.ascontiguousarray(is_device, dtype=np.uint8)
    ans = NbAnswer()
    gpus, ng = None, 0
    if devices:
        gpus = (C.c_int * len(devices))(*devices)
        ng = len(devices)
    opt = NbSolveOptions(max_batch=max_batch, engine=_ENGINES[engine], streams=_STREAMS[streams],
                         p3_parallel=p3_parallel, graph_chunk=graph_chunk, handoff=handoff)
    rc = lib().nb_solve_ex(n, planet, asteroid, *[p for _, p in keep], dev_arr.ctypes.data_as(_u8p), gpus, ng,
                           C.byref(opt), C.byref(ans))
    _check(rc, "nb_solve_ex")
    return ans.min_dist, ans.hit_time_step, ans.gravity_device_id, ans.missile_cost


NB_PHASE_WHOLE, NB_PHASE_FIRST, NB_PHASE_LAST, NB_PHASE_MIDDLE = 0, 1, 2, 3


class Sharded:
    """nb_sharded: N bodies sharded by index over the GPUs `devices` of this node, driven by this one process
    (one stream + one in-place all-gather per GPU per step: RCCL, or with exchange="copy" peer copies on the copy
    engines — the form that lets several ranks share one GPU)."""

    def __init__(self, n, devices=(0,), precision=NB_F32, G=6.674e-11, eps=1e-3, dt=60.0, overlap=False,
                 exchange="rccl", ordered_pairs=False, deadline=0.0):
        if exchange not in ("rccl", "copy", "host"):
            raise ValueError("exchange must be 'rccl', 'copy' or 'host'")
        self.n = n
        self.devices = list(devices)
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        flags = (NB_SHARDED_OVERLAP if overlap else 0) | (NB_SHARDED_COPY_EXCHANGE if exchange == "copy" else 0) | \
            (NB_SHARDED_HOST_EXCHANGE if exchange == "host" else 0) | \
            (NB_SHARDED_ORDERED_PAIRS if ordered_pairs else 0)
        rc = lib().nb_sharded_create(C.byref(self._h), devs, len(devices), n, precision, G, eps, dt, flags)
        if rc != NB_OK:
            h, self._h = self._h, C.c_void_p()
            detail = lib().nb_sharded_last_error(h).decode()
            if h:
                lib().nb_sharded_destroy(h)
            raise NBodyError(rc, "nb_sharded_create", detail)
        self.note = lib().nb_sharded_last_error(self._h).decode()  # "" or why the unordered-pair step was not taken
        if deadline:
            self.set_deadline(deadline)

    def set_deadline(self, seconds):
        """Bound every wait: a step that has not finished `seconds` after the host started waiting for it fails the call with
        NBodyError(NB_ERR_HIP, "... timed out ...") instead of blocking; 0 = wait as long as it takes."""
        self._check(lib().nb_sharded_set_deadline(self._h, float(seconds)), "nb_sharded_set_deadline")

    def _check(self, rc, where):
        if rc != NB_OK:
            raise NBodyError(rc, where, lib().nb_sharded_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().nb_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, q, v, m):
        keep = [_d(q[0]), _d(q[1]), _d(q[2]), _d(v[0]), _d(v[1]), _d(v[2]), _d(m)]
        self._check(lib().nb_sharded_set_state(self._h, *[p for _, p in keep]), "nb_sharded_set_state")

    def get_state(self):
        q, v = np.empty((3, self.n)), np.empty((3, self.n))
        ptrs = [q[k].ctypes.data_as(_dp) for k in range(3)] + [v[k].ctypes.data_as(_dp) for k in range(3)]
        self._check(lib().nb_sharded_get_state(self._h, *ptrs), "nb_sharded_get_state")
        return q, v

    def step(self, count=1):
        self._check(lib().nb_sharded_step(self._h, count), "nb_sharded_step")

    def save_state(self, path, step=0):
        """One NBODYST2 checkpoint of the whole system (atomic replace)."""
        self._check(lib().nb_sharded_save_state(self._h, os.fsencode(path), step), "nb_sharded_save_state")

    def load_state(self, path):
        """Resume from a checkpoint of the same system (n, precision, G, eps, dt must match) -> its step index."""
        step = C.c_int()
        self._check(lib().nb_sharded_load_state(self._h, os.fsencode(path), C.byref(step)), "nb_sharded_load_state")
        return step.value

    def step_timed(self, count):
        ms = C.c_double()
        self._check(lib().nb_sharded_step_timed(self._h, count, C.byref(ms)), "nb_sharded_step_timed")
        return ms.value

    def step_profiled(self, count):
        """-> (host wall ms per step, [mean GPU ms of one step's launch sequence, per rank]); count <= 1024."""
        ms = C.c_double()
        k = (C.c_float * len(self.devices))()
        self._check(lib().nb_sharded_step_profiled(self._h, count, C.byref(ms), k), "nb_sharded_step_profiled")
        return ms.value, list(k)

    def rank_info(self, rank):
        r = NbShardedRank()
        self._check(lib().nb_sharded_rank_info(self._h, rank, C.byref(r)), "nb_sharded_rank_info")
        return dict(rank=rank, device=r.device, compute_units=r.compute_units, first_target=r.first_target,
                    targets=r.targets, exchange={NB_EXCHANGE_RCCL: "rccl", NB_EXCHANGE_COPY: "copy", NB_EXCHANGE_HOST: "host"}[r.exchange],
                    comm_ranks=r.comm_ranks, comm_rank=r.comm_rank, comm_device=r.comm_device,
                    pci_bus_id=r.pci_bus_id.decode(), uuid=r.uuid.decode(), name=r.name.decode())

    def kernel_name(self):
        return lib().nb_sharded_kernel_name(self._h).decode()

    def info(self):
        p, per, r, j, w = C.c_int(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        self._check(lib().nb_sharded_info(self._h, C.byref(p), C.byref(per), C.byref(r), C.byref(j), C.byref(w)),
                    "nb_sharded_info")
        return dict(devices=p.value, targets_per_device=per.value, targets_per_lane=r.value, j_split=j.value,
                    wg_size=w.value)


def _launch_struct(src_ptr, out_ptr, n_src, tgt_off, n_tgt, eps2, dt, vel_ptr=0, pos64_ptr=0, vel64_ptr=0, acc_ptr=0,
                   acc64=False, targets_per_lane=0, j_split=0, workspace_ptr=0, workspace_bytes=0, source_path=0,
                   wg_size=0, phase=NB_PHASE_WHOLE, src_begin=0, src_end=0, tgt_ptr=0):
    return NbLaunchF32(src_ptr or None, out_ptr or None, vel_ptr or None, pos64_ptr or None, vel64_ptr or None,
                       acc_ptr or None, workspace_ptr or None, workspace_bytes, n_src, tgt_off, n_tgt, eps2, dt,
                       int(acc64), targets_per_lane, j_split, source_path, wg_size, phase, src_begin, src_end,
                       tgt_ptr or None)


def launch_f32(src_ptr, out_ptr, n_src, tgt_off, n_tgt, eps2, dt, stream, accel_only=False, **kw):
    """Raw launch on caller-owned device memory (pointers as ints, e.g. torch.Tensor.data_ptr()).
    kw: vel_ptr, pos64_ptr, vel64_ptr, acc_ptr, acc64, targets_per_lane, j_split, workspace_ptr, workspace_bytes,
    source_path, wg_size, phase, src_begin, src_end, tgt_ptr."""
    a = _launch_struct(src_ptr, out_ptr, n_src, tgt_off, n_tgt, eps2, dt, **kw)
    f = lib().nb_launch_accel_f32 if accel_only else lib().nb_launch_step_f32
    _check(f(C.byref(a), C.c_void_p(stream)), "nb_launch_accel_f32" if accel_only else "nb_launch_step_f32")


def launch_pair_forces_f32(src_ptr, n_src, tgt_off, n_tgt, eps2, stream, acc_ptr, workspace_ptr, workspace_bytes, acc64=False):
    """Several GPUs sharing the unordered pairs: this rank's partial force on ALL n_src bodies -> acc (float4 / double4)."""
    a = _launch_struct(src_ptr, 0, n_src, tgt_off, n_tgt, eps2, 0.0, acc_ptr=acc_ptr, acc64=acc64,
                       workspace_ptr=workspace_ptr, workspace_bytes=workspace_bytes)
    _check(lib().nb_launch_pair_forces_f32(C.byref(a), C.c_void_p(stream)), "nb_launch_pair_forces_f32")


def launch_kick_drift_f32(src_ptr, out_ptr, n_src, tgt_off, n_tgt, dt, stream, acc_ptr, parts=1, vel_ptr=0, pos64_ptr=0,
                          vel64_ptr=0, acc64=False):
    """Kick + drift of the rank's shard from the summed force acc[parts][n_tgt]."""
    a = _launch_struct(src_ptr, out_ptr, n_src, tgt_off, n_tgt, 1.0, dt, vel_ptr=vel_ptr, pos64_ptr=pos64_ptr,
                       vel64_ptr=vel64_ptr, acc_ptr=acc_ptr, acc64=acc64)
    _check(lib().nb_launch_kick_drift_f32(C.byref(a), parts, C.c_void_p(stream)), "nb_launch_kick_drift_f32")


def workspace_bytes_shared_pairs_f32(n_src, ranks, acc64=False):
    """Workspace of nb_launch_pair_forces_f32; 0 = the ranks cannot share the unordered pairs of this system."""
    return lib().nb_workspace_bytes_shared_pairs_f32(n_src, ranks, int(acc64))


def plan_shared_pairs_f32(n_src, ranks, acc64=False):
    """-> (superblocks per rank, workgroups per superblock, sub-launches per rank) of launch_pair_forces_f32 on the current
    device — what the kernel will do, from the library (not re-derived here)."""
    nb, wg, sub = C.c_int(), C.c_int(), C.c_int()
    _check(lib().nb_plan_shared_pairs_f32(n_src, ranks, int(acc64), C.byref(nb), C.byref(wg), C.byref(sub)),
           "nb_plan_shared_pairs_f32")
    return nb.value, wg.value, sub.value


def selftest_pair_schedule(n, n_cus=256, ranks=1, acc64=False):
    """Host-only replay of K1s' pair schedule (no GPU needed); raises NBodyError with the first inconsistency."""
    buf = C.create_string_buffer(256)
    rc = lib().nb_selftest_pair_schedule(n, n_cus, ranks, int(acc64), buf, len(buf))
    if rc != NB_OK:
        raise NBodyError(rc, "nb_selftest_pair_schedule", buf.value.decode())


def selftest_pair_schedule_within(n, workspace_bytes, n_cus=256, acc64=False):
    """The same for one GPU whose K1s workspace is limited to `workspace_bytes` (batches of superblocks)."""
    buf = C.create_string_buffer(256)
    rc = lib().nb_selftest_pair_schedule_within(n, n_cus, int(acc64), int(workspace_bytes), buf, len(buf))
    if rc != NB_OK:
        raise NBodyError(rc, "nb_selftest_pair_schedule_within", buf.value.decode())


def plan_f32(n_src, n_tgt, acc64=False, targets_per_lane=0, j_split=0, workspace_bytes=0, source_path=0, wg_size=0):
    """(targets_per_lane, j_split, wg_size) the launches will use; workspace_bytes > 0 = a workspace will be passed."""
    a = _launch_struct(1, 0, n_src, 0, n_tgt, 1.0, 1.0, acc64=acc64, targets_per_lane=targets_per_lane,
                       j_split=j_split, workspace_ptr=1 if workspace_bytes else 0, workspace_bytes=workspace_bytes,
                       source_path=source_path, wg_size=wg_size)
    r, j, w = C.c_int(), C.c_int(), C.c_int()
    _check(lib().nb_plan_f32(C.byref(a), C.byref(r), C.byref(j), C.byref(w)), "nb_plan_f32")
    return r.value, j.value, w.value


def workspace_bytes_f32(n_tgt, acc64=False):
    return lib().nb_workspace_bytes_f32(n_tgt, int(acc64))


def workspace_bytes_sym_f32(n, acc64=False):
    """Bytes that let a whole-system launch of n bodies use the symmetric kernel K1s; 0 = not applicable to this n."""
    return lib().nb_workspace_bytes_sym_f32(n, int(acc64))


def kernel_name_f32(n_src, n_tgt, acc64=False, targets_per_lane=0, j_split=0, workspace_bytes=0, accel_only=False,
                    source_path=0, wg_size=0):
    a = _launch_struct(1, 0, n_src, 0, n_tgt, 1.0, 1.0, acc64=acc64, targets_per_lane=targets_per_lane,
                       j_split=j_split, workspace_ptr=1 if workspace_bytes else 0, workspace_bytes=workspace_bytes,
                       source_path=source_path, wg_size=wg_size)
    return lib().nb_kernel_name_f32(C.byref(a), int(accel_only)).decode()
