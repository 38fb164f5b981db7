#!/usr/bin/env python3
"""bench/k1_small_n_jsplit.py — K1 (every ordered pair) for systems below K1s' threshold: the fused step with the source-slice count
forced (any value, not only the powers of two plan_f32 picks), by system size.  256-thread workgroups of 1024 targets run two per CU:
does a slice count that fills whole rounds of 512 workgroups beat the plan's?"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

for n in (12288, 16384, 20480, 24576, 28672, 32768, 34816):
    pos, vel = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    out = torch.zeros_like(src)
    v = torch.from_numpy(vel).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(66 * n * 16, dtype=torch.uint8, device="cuda")  # room for 64 slices in one launch
    auto = c.plan_f32(n, n, False, 0, 0, ws.numel(), 2)
    res = {}
    blocks = -(-n // 1024)
    cands = sorted({0, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64} | {max(2, 512 // blocks), max(2, 1024 // blocks), max(2, 768 // blocks)})
    for js in cands:
        if js > n // 256 // 2 and js:
            continue
        step = lambda: c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, syn.DT, stream, vel_ptr=v.data_ptr(),  # noqa: E731
                                    source_path=2, j_split=js, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
        best = 1e9
        for rnd in range(3):
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                step()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        res[js] = best
    line = "  ".join(f"{('auto=' + str(auto[1])) if js == 0 else js}: {t:.4f}" for js, t in res.items())
    b = min(res, key=res.get)
    print(f"n = {n:6d} ({blocks} target blocks)  ms/step by source slices  {line}   best {b or 'auto'} ({(res[0] / res[b] - 1) * 100:.0f} % over auto)", flush=True)
