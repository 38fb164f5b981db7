#!/usr/bin/env python3
"""bench/sym_min_phases_ab.py LIB_A LIB_B ... — K1s / K1s-f64 for SMALL systems with two builds of the library that differ in the least number of tile
phases a workgroup executes (-DNB_SYM_MIN_PHASES=8, the product until round 5, against 4 / 2; both with -DNB_SYM_MIN_N=8192 -DNB_SYM64_MIN_SB=4 so
that the kernels can be asked for below their thresholds).  Per size and build: workgroups per superblock the plan picks, ms per fused step (best
of 3), max |K1s - K1| / max |a| over ALL bodies (fp32), and for fp64 the step time of an NB_F64 context forced onto K1s-f64 with 16 rows against the
oracle.  K1 (every ordered pair) once, with the first build, as the yardstick."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402
from oracle import oracle as O  # noqa: E402  (a measurement tool, not the product)

LIBS = sys.argv[1:]
SIZES = (8192, 12288, 16384, 20480, 24576 - 77, 28672, 32768, 36864, 40960, 45056 + 5, 49152, 57344, 65536, 81920, 98304, 131072)


def timed(step, n):
    best = 1e9
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(10, int(2e10 / (n * n)))
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def f32_case(n, sp):
    pos, vel = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    out = torch.zeros_like(src)
    v = torch.from_numpy(vel).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    need = c.workspace_bytes_sym_f32(n) if sp == 3 else c.workspace_bytes_f32(n) // 18 * 66
    if need <= 0:
        return None
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ms = timed(lambda: c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, syn.DT, stream, vel_ptr=v.data_ptr(),
                                    source_path=sp, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel()), n)
    a = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    c.launch_f32(src.data_ptr(), 0, n, 0, n, syn.EPS ** 2, syn.DT, stream, accel_only=True, acc_ptr=a.data_ptr(), source_path=sp,
                 workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
    torch.cuda.synchronize()
    return ms, c.plan_f32(n, n, False, 0, 0, need, sp)[1], a[:, :3].double()


def f64_case(n):
    q, v, m = syn.bodies(n)
    with c.Context(n, c.NB_F64, 0, G=syn.G, eps=syn.EPS, dt=syn.DT, f64_large_min=1024) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 3)
        k = max(10, int(5e9 / (n * n)))
        ms = min(ctx.step_timed(4 + r * k, k) for r in range(3))
        ctx.set_state(q, v, m)
        a = ctx.accel(1)
    rows = np.unique(np.linspace(0, n - 1, 16).astype(np.int64))
    r, s_ = O.accel_rows_at(q, m, syn.G, syn.EPS, rows, want_abs=True)
    return ms, float((np.abs(a[:, rows] - r).max(axis=0) / s_).max())


for n in SIZES:
    line = f"n = {n:6d} ({-(-n // 4096):2d} x 4096 | {-(-n // 2048):2d} x 2048)"
    ref = None
    for k, lib in enumerate(LIBS):
        with c.use_library(lib):
            if k == 0:
                ms1, js1, ref = f32_case(n, 2)
                line += f"  K1 {ms1:.4f} ms"
            r = f32_case(n, 3)
            tag = os.path.basename(os.path.dirname(lib))
            if r:
                ms, chunks, a = r
                err = ((a - ref).abs().max() / ref.abs().max()).item()
                line += f"  | {tag}: K1s {ms:.4f} ms ({chunks} wg/sb, {20 * n * (n - 1) / (ms * 1e-3) / 157.3e12:.3f} of peak, vs K1 {err:.1e})"
            if n <= 65536:
                ms64, e64 = f64_case(n)
                line += f"  f64 {ms64:.4f} ms ({20 * n * (n - 1) / (ms64 * 1e-3) / 78.6e12:.3f}, oracle {e64:.1e})"
    print(line, flush=True)
