#!/usr/bin/env python3
"""bench/workspace_cap_ab.py — what a limit on K1s' pair-slot workspace costs (nb_config.flags, NB_CFG_WORKSPACE_GIB): the same
fp32 system stepped by contexts with no limit (one launch, the fastest shape) and with tighter and tighter limits (batches of
superblocks whose reducers add up a running force), alternating on one GPU; device memory taken, ms per step, and the
accelerations against the unlimited context's (same pairs, other summation cuts: fp32 rounding).  Round 5."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402


def used():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e9


def main():
    torch.cuda.init()
    for n, caps, steps in ((1 << 20, (0, 1), 6), (1 << 22, (0, 8, 3), 2)):
        q, v, m = syn.bodies(n)
        ref = None
        for rnd in range(2):
            for cap in caps:
                u0 = used()
                with c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=syn.DT, workspace_gib=cap) as ctx:
                    ctx.set_state(q, v, m)
                    ctx.step(1, 1)
                    ms = ctx.step_timed(2, steps)
                    mem = used() - u0
                    name, note = ctx.kernel_name(), ctx.last_error()
                    a = ctx.accel(2 + steps) if rnd == 0 else None
                if a is not None:
                    if ref is None:
                        ref = a
                    err = float(np.abs(a - ref).max() / np.abs(ref).max())
                else:
                    err = float("nan")
                print(f"n = 2^{n.bit_length() - 1}  limit {cap or 'none':>4} GiB  {mem:6.2f} GB on the device  {ms:9.3f} ms/step  {name}  "
                      f"max|a - a_unlimited| / max|a| = {err:.1e}  {note[:90]}", flush=True)


if __name__ == "__main__":
    main()
