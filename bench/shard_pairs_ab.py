#!/usr/bin/env python3
"""Time ONE rank's step of a P-way run on a single GPU, both forms, interleaved: every ordered pair of the rank's targets
(K1: nb_launch_step_f32, targets = shard `rank`, sources = all N) against the rank's share of the UNORDERED pairs of the
system (K1s: nb_launch_pair_forces_f32 -> its partial force on all N bodies; the reduce-scatter and the kick-drift of the real
step are not part of either timing).   python bench/shard_pairs_ab.py N P [rank ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi, synthetic  # noqa: E402
from nbody_amd.distributed import workspace_bytes  # noqa: E402

n, p = int(sys.argv[1]), int(sys.argv[2])
ranks = [int(x) for x in sys.argv[3:]] or [0, p - 1]
per = n // p
pos, vel = synthetic.body4_f32(n)
src = torch.from_numpy(pos).cuda()
out = torch.zeros_like(src)
stream = torch.cuda.current_stream().cuda_stream
ws1 = torch.empty(workspace_bytes(n, per), dtype=torch.uint8, device="cuda")
ws2 = torch.empty(capi.workspace_bytes_shared_pairs_f32(n, p), dtype=torch.uint8, device="cuda")
fpart = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
reps = max(3, int(1e12 / (per * n)))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(2):
    for r in ranks:
        v = torch.from_numpy(vel[r * per:(r + 1) * per].copy()).cuda()
        ms1 = timed(lambda: capi.launch_f32(src.data_ptr(), out.data_ptr(), n, r * per, per, synthetic.EPS ** 2, synthetic.DT,
                                            stream, vel_ptr=v.data_ptr(), workspace_ptr=ws1.data_ptr(), workspace_bytes=ws1.numel()))
        ms2 = timed(lambda: capi.launch_pair_forces_f32(src.data_ptr(), n, r * per, per, synthetic.EPS ** 2, stream,
                                                         fpart.data_ptr(), ws2.data_ptr(), ws2.numel()))
        f = lambda ms: 20 * per * (n - 1) / (ms * 1e-3) / 157.3e12  # noqa: E731
        print(f"round {rnd + 1} N={n} P={p} rank {r}: ordered pairs (K1) {ms1:.3f} ms = {f(ms1):.4f} of peak; shared unordered "
              f"pairs (K1s) {ms2:.3f} ms = {f(ms2):.4f}; slots workspace {ws2.numel() / 1e9:.2f} GB", flush=True)
