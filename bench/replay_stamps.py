#!/usr/bin/env python3
"""Per-launch timing of the per-step fp64 engine INSIDE a replayed hipGraph, from the kernel's own clock stamps
(the instrumented build libnbody_amd_stamps.so, nb_enable_step_stamps: 100 MHz GPU wall clock at kernel entry and after
the last store of workgroup 0; the hook itself lengthens the step by 7-13 %, so read the SPLIT, not the absolute period).  rocprofv3's
kernel tracing crashes inside hipGraphLaunch on this image (profiles/r03_graph_trace_limit.txt), so this is how the replay
path is measured: duration of one step launch, gap to the next node of the graph (the dependent-kernel boundary), and the
period — against the same launches issued eagerly from the host.
    python bench/replay_stamps.py [b200 b512 b1024]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c  # noqa: E402
from oracle import oracle as O  # noqa: E402  (input parsing only)

TICK_US = 0.01  # wall_clock64: 100 MHz
STEPS, CHUNK = 20000, 1000


def stats(x):
    return f"median {np.median(x):6.2f}  mean {x.mean():6.2f}  p5 {np.percentile(x, 5):6.2f}  p95 {np.percentile(x, 95):6.2f}"


c.use_library(c.stamps_library_path()).__enter__()  # the instrumented build for the whole run of this tool

for case in sys.argv[1:] or ["b200", "b512", "b1024"]:
    s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", case + ".in"))
    for mode, flags in (("graph replay", 0), ("eager launches", c.NB_SCN_EAGER)):
        with c.Context(s.n) as x:
            x.set_state(s.q, s.v, s.m, s.is_device)
            x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=200, engine=1)  # warm-up
            x.enable_step_stamps(CHUNK)
            t0 = time.perf_counter()
            r = x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, first_step=200, last_step=200 + STEPS, engine=1,
                               flags=flags, graph_chunk=CHUNK)
            wall = (time.perf_counter() - t0) / STEPS * 1e6
            st = x.read_step_stamps(CHUNK).astype(np.int64)
        assert r["steps_done"] == 200 + STEPS
        if not (st[1:, 1] > 0).any():
            print(f"{case} {mode}: no exit stamps; first rows of the buffer:\n{st[:6]}", flush=True)
            continue
        ok = st[1:, 1] > 0                      # slot 0 was overwritten by the monitor-only launch after the last step
        ent, ext = st[1:, 0][ok], st[1:, 1][ok]
        dur = (ext - ent) * TICK_US
        # consecutive slots are consecutive launches (of the last replay / the last CHUNK eager steps)
        gap = (st[2:, 0] - st[1:-1, 1])[ok[:-1] & ok[1:]] * TICK_US
        per = (st[2:, 0] - st[1:-1, 0])[ok[:-1] & ok[1:]] * TICK_US
        gap, per = gap[(gap > -50) & (gap < 50)], per[(per > 0) & (per < 100)]  # (the ring wraps once in eager mode)
        print(f"{case} n={s.n:5d} {mode:14s}: host wall {wall:5.2f} us/step | kernel (entry -> last store of workgroup 0) "
              f"us: {stats(dur)} | gap to the next launch us: {stats(gap)} | period us: {stats(per)}", flush=True)
