#!/usr/bin/env python3
"""bench/chunk_sweep.py [n ...] — K1s' fused step through the raw launch (source_path 3) with the number of workgroups per superblock
forced (j_split), alternating, per system size: is the library's choice (sym_choose_chunks: fill whole rounds of workgroups, 0.4 % per
extra chunk) still the fastest now that the work is cut at tile-phase granularity?"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [1 << 17, 196608, 1 << 18, 1 << 19, 1 << 20]
for n in sizes:
    pos, vel = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    out = torch.zeros_like(src)
    v = torch.from_numpy(vel).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(int(c.workspace_bytes_sym_f32(n) * 3) + (1 << 28), dtype=torch.uint8, device="cuda")  # room for many chunks
    reps = max(4, int(2e12 / (n * n)))
    auto = c.plan_f32(n, n, False, 0, 0, ws.numel(), 3)[1]
    res = {}
    for rnd in range(2):
        for ch in (0, 1, 2, 4, 8, 16, 32):
            try:
                step = lambda: c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, syn.DT, stream, vel_ptr=v.data_ptr(),  # noqa: E731
                                            source_path=3, j_split=ch, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
                step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(ch, []).append(e0.elapsed_time(e1) / reps)
            except c.NBodyError as e:
                res.setdefault(ch, []).append(float("nan"))
    line = "  ".join(f"{('auto=' + str(auto)) if ch == 0 else ch}: {min(t):8.3f}" for ch, t in res.items())
    best = min((min(t), ch) for ch, t in res.items() if min(t) == min(t))
    print(f"n = {n:8d} (B = {-(-n // 4096):3d})  ms/step by workgroups per superblock  {line}   best {best[1] or 'auto'}", flush=True)
