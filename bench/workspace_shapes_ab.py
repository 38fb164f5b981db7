#!/usr/bin/env python3
"""bench/workspace_shapes_ab.py LIB_A LIB_B — two builds of the library, alternating on one GPU: K1s' default workspace shapes of
round 5 (linear in n: one launch up to 2 GiB of slots, beyond that batches within 720 B per body) against rounds 1-4's (one
launch up to 32 GiB, batches within 64 GiB; `make LIB=bench/ab/r04shapes/libnbody_amd.so EXTRA="-DNB_SYM_WHOLE_GIB=32
-DNB_SYM_BATCH_FLOOR_GIB=64" lib`).  Whole systems through nb_create / nb_step_timed at n = 2^21, 2^22, 2^23, and ONE rank's
share of BASELINE configs[3] and configs[4] through nb_launch_pair_forces_f32.  Plain ctypes: both builds in one process."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi, synthetic as syn  # noqa: E402


def used():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e9


def main():
    torch.cuda.init()
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    for n, steps in ((1 << 21, 3), (1 << 22, 2), (1 << 23, 1)):
        q, v, m = syn.bodies(n)
        for rnd in range(2):
            for path in libs:
                with capi.use_library(path):
                    u0 = used()
                    with capi.Context(n, capi.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
                        ctx.set_state(q, v, m)
                        ctx.step(1, 1)
                        ms = ctx.step_timed(2, steps)
                        mem = used() - u0
                print(f"whole system  n = 2^{n.bit_length() - 1}  {os.path.relpath(path, ROOT):45s} {mem:6.2f} GB on the device  {ms:10.3f} ms/step",
                      flush=True)
        del q, v, m
    for n, ranks, acc64, reps in ((1 << 22, 8, False, 3), (1 << 24, 8, True, 1)):
        pos, _ = syn.body4_f32(n)
        src = torch.from_numpy(pos).cuda()
        per = n // ranks
        part = torch.empty((n, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for rnd in range(2):
            for path in libs:
                with capi.use_library(path):
                    nbytes = capi.workspace_bytes_shared_pairs_f32(n, ranks, acc64)
                    plan = capi.plan_shared_pairs_f32(n, ranks, acc64)
                    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                    run = lambda: capi.launch_pair_forces_f32(src.data_ptr(), n, 3 * per, per, syn.EPS ** 2, stream, part.data_ptr(),  # noqa: E731
                                                              ws.data_ptr(), ws.numel(), acc64=acc64)
                    run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / reps
                    del ws
                    torch.cuda.empty_cache()
                print(f"rank 3 of {ranks}   n = 2^{n.bit_length() - 1}  {os.path.relpath(path, ROOT):45s} {nbytes / 1e9:6.2f} GB of slots  {ms:10.3f} ms  "
                      f"plan (superblocks, workgroups each, sub-launches) {plan}  = {20 * per * (n - 1) / (ms * 1e-3) / 157.3e12:.4f} of peak", flush=True)
        del src, part


if __name__ == "__main__":
    main()
