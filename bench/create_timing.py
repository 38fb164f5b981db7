#!/usr/bin/env python3
"""bench/create_timing.py [lib ...] — wall time and device memory of nb_create, then of the first and second nb_step, for fp32
contexts of 2^20 and 2^22 bodies, per library build (default: the in-tree one).  Round 5 moved the pair-slot workspace of the
unordered-pair kernel K1s (1.7 GB at 2^20, 26 GB at 2^22) from nb_create to the first step that uses it:
run with the round-4 build beside the current one (`bench/ab/kahan/libnbody_amd.so`, built before the change) to see both.
Plain ctypes on purpose: an older build lacks symbols the package's binding insists on."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch  # device-wide memory figures (hipMemGetInfo), and it brings the HIP runtime both builds bind to

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Cfg(C.Structure):
    _fields_ = [("n", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32), ("f64_large_min", C.c_int32),
                ("f64_split", C.c_int32), ("flags", C.c_int32), ("G", C.c_double), ("eps", C.c_double), ("dt", C.c_double)]


def used():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e9


def run(path, n):
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    cfg = Cfg()
    L.nb_config_default(C.byref(cfg))
    cfg.n, cfg.precision, cfg.dt = n, 1, 1e-4
    rng = np.random.default_rng(1)
    q, v = rng.uniform(-1, 1, (3, n)), rng.uniform(-1e-3, 1e-3, (3, n))
    m = np.full(n, 1.0 / (n * cfg.G))
    ptr = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    h = C.c_void_p()
    u0 = used()
    t0 = time.perf_counter()
    rc = L.nb_create(C.byref(h), C.byref(cfg))
    t_create = time.perf_counter() - t0
    u1 = used()
    assert rc == 0, rc
    assert L.nb_set_state(h, ptr(q[0]), ptr(q[1]), ptr(q[2]), ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(m), None) == 0
    t0 = time.perf_counter()
    assert L.nb_step(h, 1, 1) == 0
    t_first = time.perf_counter() - t0
    u2 = used()
    t0 = time.perf_counter()
    assert L.nb_step(h, 2, 1) == 0
    t_second = time.perf_counter() - t0
    L.nb_destroy(h)
    print(f"{os.path.relpath(path, ROOT):48s} n=2^{n.bit_length() - 1}: nb_create {t_create * 1e3:8.1f} ms (+{u1 - u0:5.2f} GB)   first nb_step "
          f"{t_first * 1e3:8.1f} ms (+{u2 - u1:5.2f} GB)   second nb_step {t_second * 1e3:8.1f} ms", flush=True)


if __name__ == "__main__":
    torch.cuda.init()
    libs = sys.argv[1:] or [os.path.join(ROOT, "nthu_ipc_nbody-simulation_amd", "libnbody_amd.so")]
    for n in (1 << 20, 1 << 22):
        for rnd in range(2):
            for lib in libs:
                run(os.path.abspath(lib), n)
