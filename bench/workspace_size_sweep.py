#!/usr/bin/env python3
"""bench/workspace_size_sweep.py [n=1048576] [reps=6] — one system, one GPU, the fused K1s step through the raw launch
(nb_launch_step_f32, source_path 3) with workspaces of different sizes, alternating: the full one (one launch, a slot per
round) and smaller ones (batches of superblocks).  Which shape is fastest at the metric's N?"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pos, vel = syn.body4_f32(n)
src = torch.from_numpy(pos).cuda()
out = torch.zeros_like(src)
v = torch.from_numpy(vel).cuda()
stream = torch.cuda.current_stream().cuda_stream
full = c.workspace_bytes_sym_f32(n)
sizes = [full] + [int(f * 12 * n) for f in (140, 80, 62, 52, 46)]
bufs = {b: torch.empty(b, dtype=torch.uint8, device="cuda") for b in sizes}


def step(ws):
    c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, syn.DT, stream, vel_ptr=v.data_ptr(), source_path=0,
                 workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())


for rnd in range(3):
    for b in sizes:
        ws = bufs[b]
        name = c.kernel_name_f32(n, n, workspace_bytes=b)
        step(ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step(ws)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"round {rnd + 1}  n = {n}  workspace {b / 1e9:5.2f} GB ({b / (12 * n):5.1f} slots)  {ms:9.3f} ms/step  "
              f"{20 * n * (n - 1) / (ms * 1e-3) / 157.3e12:.4f} of peak  {name}", flush=True)
