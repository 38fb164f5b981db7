#!/usr/bin/env python3
"""Time ONE rank's force+kick-drift launch of an index-sharded run on a single GPU: targets = the first N/P bodies,
sources = all N (what rank 0 of a P-GPU job executes per step, without the all-gather).  Interleaved A/B of launch
shapes on one device.   python bench/shard_kernel_ab.py N P "wg tpl js [slots]" ["wg tpl js [slots]" ...]
slots = partial-sum slots of the workspace handed to the launch (default 16: js/16 launches + reducers per step; up to 64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402
from nbody_amd import capi, synthetic  # noqa: E402

n, p = int(sys.argv[1]), int(sys.argv[2])
cfgs = [tuple(int(x) for x in c.split()) for c in sys.argv[3:]] or [(0, 0, 0)]
cfgs = [c if len(c) == 4 else c + (16,) for c in cfgs]
n_tgt = n // p
pos, vel = synthetic.body4_f32(n)
src = torch.from_numpy(pos).cuda()
out = torch.zeros_like(src)
v = torch.from_numpy(vel[:n_tgt].copy()).cuda()
ws_all = {k: torch.empty((k + 2) * n_tgt * 16, dtype=torch.uint8, device="cuda") for k in {c[3] for c in cfgs}}
stream = torch.cuda.current_stream().cuda_stream
reps = max(3, int(2e12 / (n_tgt * n)))
for rnd in range(2):
    for wg, tpl, js, slots in cfgs:
        ws = ws_all[slots]
        kw = dict(vel_ptr=v.data_ptr(), targets_per_lane=tpl, j_split=js, wg_size=wg, workspace_ptr=ws.data_ptr(),
                  workspace_bytes=ws.numel())
        capi.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n_tgt, synthetic.EPS ** 2, synthetic.DT, stream, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            capi.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n_tgt, synthetic.EPS ** 2, synthetic.DT, stream, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        plan = capi.plan_f32(n, n_tgt, targets_per_lane=tpl, j_split=js, wg_size=wg, workspace_bytes=ws.numel())
        rate = n_tgt * (n - 1) / (ms * 1e-3)
        print(f"round {rnd + 1} N={n} P={p} asked(wg,R,js,slots)=({wg},{tpl},{js},{slots}) plan(R,js,wg)={plan}: {ms:.3f} ms/step/rank, "
              f"{rate:.4e} pairs/s per rank ({100 * rate * 20 / 157.3e12:.2f}% of peak), x{p} = {rate * p:.4e}")
