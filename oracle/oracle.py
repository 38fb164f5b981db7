"""ctypes door to the CPU oracle (oracle/nbody_oracle.c) and, when built, the real reference (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
REF = os.path.join(HERE, "_ref")

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class OrcParams(C.Structure):
    _fields_ = [("n_steps", C.c_int), ("dt", C.c_double), ("eps", C.c_double), ("G", C.c_double),
                ("planet_radius", C.c_double), ("missile_speed", C.c_double)]


class OrcSystem(C.Structure):
    _fields_ = [("n", C.c_int), ("planet", C.c_int), ("asteroid", C.c_int),
                ("qx", _dp), ("qy", _dp), ("qz", _dp), ("vx", _dp), ("vy", _dp), ("vz", _dp), ("m", _dp),
                ("is_device", _u8p)]


class OrcResult(C.Structure):
    _fields_ = [("min_dist", C.c_double), ("hit_time_step", C.c_int), ("gravity_device_id", C.c_int),
                ("missile_cost", C.c_double)]


class OrcP3Detail(C.Structure):
    _fields_ = [("device", C.c_int), ("arrival_step", C.c_int), ("feasible", C.c_int), ("fail_step", C.c_int),
                ("cost", C.c_double)]


def build(quiet=True):
    """Compile the oracle (and the reference under oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL if quiet else None)


def _dptr(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _u8ptr(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(_u8p)


_libs = {}


def lib(omp=False):
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if name not in _libs:
        path = os.path.join(BUILD, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_gravity_device_mass.restype = C.c_double
        L.orc_gravity_device_mass.argtypes = [C.c_double, C.c_double]
        L.orc_missile_cost.restype = C.c_double
        L.orc_missile_cost.argtypes = [C.c_double]
        L.orc_effective_mass.argtypes = [C.c_int, C.c_int, _dp, _u8p, C.c_double, _dp]
        L.orc_accel_rows.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int,
                                     _dp, _dp, _dp, _dp]
        L.orc_accel_rows_at.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_int,
                                        _dp, _dp, _dp, _dp]
        L.orc_run_step.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p,
                                   C.POINTER(OrcParams), _dp]
        L.orc_read_input.argtypes = [C.c_char_p, C.POINTER(OrcSystem)]
        L.orc_system_free.argtypes = [C.POINTER(OrcSystem)]
        L.orc_problem1.restype = C.c_double
        L.orc_problem1.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcParams)]
        L.orc_problem23.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcParams), C.POINTER(OrcResult),
                                    C.POINTER(OrcP3Detail), C.c_int]
        L.orc_problem3_from_zero.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcParams), C.c_int, _dp,
                                             C.POINTER(C.c_int)]
        L.orc_solve_file.argtypes = [C.c_char_p, C.c_char_p]
        _libs[name] = L
    return _libs[name]


def reference_params(omp=False):
    return OrcParams.in_dll(lib(omp), "ORC_REFERENCE_PARAMS")


def make_params(n_steps=200000, dt=60.0, eps=1e-3, G=6.674e-11, planet_radius=1e7, missile_speed=1e6):
    return OrcParams(n_steps, dt, eps, G, planet_radius, missile_speed)


class System:
    """Host-side SoA state, the oracle's view of `read_input` (samples/nbody.cc:22-39)."""

    def __init__(self, n, planet=0, asteroid=1):
        self.n, self.planet, self.asteroid = n, planet, asteroid
        self.q = np.zeros((3, n))
        self.v = np.zeros((3, n))
        self.m = np.zeros(n)
        self.is_device = np.zeros(n, dtype=np.uint8)

    def copy(self):
        s = System(self.n, self.planet, self.asteroid)
        s.q[:], s.v[:], s.m[:], s.is_device[:] = self.q, self.v, self.m, self.is_device
        return s

    def c_struct(self):
        return OrcSystem(self.n, self.planet, self.asteroid, _dptr(self.q[0]), _dptr(self.q[1]), _dptr(self.q[2]),
                         _dptr(self.v[0]), _dptr(self.v[1]), _dptr(self.v[2]), _dptr(self.m),
                         _u8ptr(self.is_device))


def read_input(path):
    raw = OrcSystem()
    rc = lib().orc_read_input(os.fsencode(path), C.byref(raw))
    if rc:
        raise OSError(f"orc_read_input({path}) -> {rc}")
    s = System(raw.n, raw.planet, raw.asteroid)
    n = raw.n
    for k, p in enumerate((raw.qx, raw.qy, raw.qz)):
        s.q[k] = np.ctypeslib.as_array(p, (n,))
    for k, p in enumerate((raw.vx, raw.vy, raw.vz)):
        s.v[k] = np.ctypeslib.as_array(p, (n,))
    s.m[:] = np.ctypeslib.as_array(raw.m, (n,))
    s.is_device[:] = np.ctypeslib.as_array(raw.is_device, (n,))
    lib().orc_system_free(C.byref(raw))
    return s


def effective_mass(step, m, is_device, dt):
    me = np.empty_like(m)
    lib().orc_effective_mass(step, len(m), _dptr(m), _u8ptr(is_device), dt, _dptr(me))
    return me


def accel_rows(q, m_eff, G, eps, i0=0, i1=None, want_abs=False, omp=True):
    """Accelerations of rows [i0,i1) in fp64, reference arithmetic (samples/nbody.cc:56-74)."""
    n = q.shape[1]
    i1 = n if i1 is None else i1
    a = np.empty((3, i1 - i0))
    ab = np.empty(i1 - i0) if want_abs else None
    lib(omp).orc_accel_rows(n, _dptr(q[0]), _dptr(q[1]), _dptr(q[2]), _dptr(m_eff), G, eps, i0, i1,
                            _dptr(a[0]), _dptr(a[1]), _dptr(a[2]), _dptr(ab) if want_abs else None)
    return (a, ab) if want_abs else a


def accel_rows_at(q, m_eff, G, eps, rows, want_abs=False, omp=True):
    """Accelerations of the target rows listed in `rows` (any order), fp64, reference arithmetic — row r of the result belongs
    to rows[r]; every bit as accel_rows gives for that row, all rows in one call (OpenMP over the list)."""
    n = q.shape[1]
    idx = np.ascontiguousarray(rows, dtype=np.int32)
    assert idx.ndim == 1 and (idx >= 0).all() and (idx < n).all()
    a = np.empty((3, len(idx)))
    ab = np.empty(len(idx)) if want_abs else None
    lib(omp).orc_accel_rows_at(n, _dptr(q[0]), _dptr(q[1]), _dptr(q[2]), _dptr(m_eff), G, eps,
                               idx.ctypes.data_as(C.POINTER(C.c_int)), len(idx), _dptr(a[0]), _dptr(a[1]), _dptr(a[2]),
                               _dptr(ab) if want_abs else None)
    return (a, ab) if want_abs else a


def run_steps(s, first_step, count, params=None, omp=False):
    """In-place `count` calls of run_step with indices first_step.. (samples/nbody.cc:51-89)."""
    p = params or reference_params(omp)
    scratch = np.empty(4 * s.n)
    L = lib(omp)
    for k in range(count):
        L.orc_run_step(first_step + k, s.n, _dptr(s.q[0]), _dptr(s.q[1]), _dptr(s.q[2]), _dptr(s.v[0]),
                       _dptr(s.v[1]), _dptr(s.v[2]), _dptr(s.m), _u8ptr(s.is_device), C.byref(p), _dptr(scratch))
    return s


def problem1(s, params=None, omp=False):
    p = params or reference_params(omp)
    cs = s.c_struct()
    return lib(omp).orc_problem1(C.byref(cs), C.byref(p))


def problem23(s, params=None, omp=False, max_detail=8):
    p = params or reference_params(omp)
    cs = s.c_struct()
    r = OrcResult()
    det = (OrcP3Detail * max_detail)()
    lib(omp).orc_problem23(C.byref(cs), C.byref(p), C.byref(r), det, max_detail)
    details = [dict(device=d.device, arrival_step=d.arrival_step, feasible=bool(d.feasible),
                    fail_step=d.fail_step, cost=d.cost) for d in det if d.device >= 0]
    return r, details


def problem3_from_zero(s, device, params=None, omp=False):
    p = params or reference_params(omp)
    cs = s.c_struct()
    cost = C.c_double()
    arr = C.c_int()
    ok = lib(omp).orc_problem3_from_zero(C.byref(cs), C.byref(p), device, C.byref(cost), C.byref(arr))
    return bool(ok), cost.value, arr.value


def solve_file(in_path, out_path, omp=False):
    rc = lib(omp).orc_solve_file(os.fsencode(in_path), os.fsencode(out_path))
    if rc:
        raise OSError(f"orc_solve_file -> {rc}")


# ---------------------------------------------------------------- the real reference (oracle/_ref)

def have_reference():
    return os.path.exists(os.path.join(REF, "libnbody_ref.so"))


_ref = None


def reflib():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(REF, "libnbody_ref.so"))
        L.ref_run_steps.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _u8p]
        L.ref_gravity_device_mass.restype = C.c_double
        L.ref_gravity_device_mass.argtypes = [C.c_double, C.c_double]
        _ref = L
    return _ref


def ref_run_steps(s, first_step, count):
    """The reference's own run_step (samples/nbody.cc:51-89 compiled in place), `count` times, in place."""
    reflib().ref_run_steps(first_step, count, s.n, _dptr(s.q[0]), _dptr(s.q[1]), _dptr(s.q[2]), _dptr(s.v[0]),
                           _dptr(s.v[1]), _dptr(s.v[2]), _dptr(s.m), _u8ptr(s.is_device))
    return s
