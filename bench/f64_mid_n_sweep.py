#!/usr/bin/env python3
"""bench/f64_mid_n_sweep.py — plain fp64 steps (NB_F64 contexts, nb_step) of whole systems between the testcase sizes (<= 1024) and K1s-f64's
threshold (32768): the default plan against K2 with every `f64_split` forced and against K1-f64 forced from that size (`f64_large_min`).
ms per step (HIP events over 200 / 50 steps, best of 3), pairs/s and the fraction of the 78.6 TFLOP/s fp64 vector peak at 20 flop per pair."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

LIB = None
if len(sys.argv) > 2 and sys.argv[1] == "--lib":  # an A/B build, e.g. make LIB=bench/ab/sym64min/libnbody_amd.so EXTRA=-DNB_SYM64_MIN_SB=4 lib
    LIB = sys.argv[2]
    del sys.argv[1:3]
SIZES = [int(a) for a in sys.argv[1:]] or [1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 20480, 24576, 28672, 32768 - 256, 32768, 49152]


def timed(n, **kw):
    q, v, m = syn.bodies(n)
    try:
        with c.Context(n, c.NB_F64, 0, G=syn.G, eps=syn.EPS, dt=syn.DT, **kw) as ctx:
            ctx.set_state(q, v, m)
            k = 200 if n <= 8192 else 50
            ctx.step(1, 5)
            return min(ctx.step_timed(6 + r * k, k) for r in range(3))
    except c.NBodyError as e:
        return float("nan")


def forced_large_error(n):
    """the K1-f64 / K1s-f64 context's accelerations of 16 rows against the oracle, relative to sum_j |a_ij|"""
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from oracle import oracle as O
    q, v, m = syn.bodies(n)
    with c.Context(n, c.NB_F64, 0, G=syn.G, eps=syn.EPS, dt=syn.DT, f64_large_min=1024) as ctx:
        ctx.set_state(q, v, m)
        a = ctx.accel(1)
    rows = np.unique(np.linspace(0, n - 1, 16).astype(np.int64))
    r, s_ = O.accel_rows_at(q, m, syn.G, syn.EPS, rows, want_abs=True)
    return float((np.abs(a[:, rows] - r).max(axis=0) / s_).max())


ctx_lib = c.use_library(LIB) if LIB else __import__("contextlib").nullcontext()
ctx_lib.__enter__()
for n in SIZES:
    res = {"auto": timed(n)}
    for s in (1, 2, 4, 8, 16, 32, 64):
        res[f"K2/S={s}"] = timed(n, f64_split=s, f64_large_min=1 << 30)
    res["K1-f64"] = timed(n, f64_large_min=1024)
    res["auto"] = min(res["auto"], timed(n))  # (again at the end: the first context of a size also pays the clock's ramp)
    best = min((k for k in res if k != "auto" and res[k] == res[k]), key=res.get)
    pk = lambda ms: n * (n - 1) / (ms * 1e-3) * 20 / 78.6e12  # noqa: E731
    print(f"n = {n:6d}  auto {res['auto']:.4f} ms ({pk(res['auto']):.3f} of peak)  best {best} {res[best]:.4f} ms ({pk(res[best]):.3f}; auto "
          f"{100 * (res['auto'] / res[best] - 1):+.0f} %)   " + "  ".join(f"{k}: {t:.4f}" for k, t in res.items() if k != "auto")
          + f"   forced-large oracle err {forced_large_error(n):.1e}", flush=True)
