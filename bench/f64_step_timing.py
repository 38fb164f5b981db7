#!/usr/bin/env python3
"""Per-step GPU time of the fp64 testcase path (K2) and host<->HBM transfer cost of the ABI (run on the GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402
from nbody_amd import capi, host, synthetic  # noqa: E402

for case in ("b20", "b50", "b100", "b200", "b512", "b1024"):
    n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, types = host.read_input(
        os.path.join(ROOT, "tests/golden/testcases", f"{case}.in"))
    dev = np.array([t == "device" for t in types], dtype=np.uint8)
    with capi.Context(n) as ctx:
        ctx.set_state(np.stack([qx, qy, qz]), np.stack([vx, vy, vz]), m, dev)
        ctx.step(1, 200)
        ms = ctx.step_timed(201, 5000)
        t0 = time.perf_counter()
        r = ctx.run_scenario(capi.NB_SCN_MIN_DIST, planet, asteroid, first_step=0, last_step=20000, engine=1)
        wall = time.perf_counter() - t0
        k3 = ""
        if n <= 128:
            ctx.run_scenario(capi.NB_SCN_MIN_DIST, planet, asteroid, first_step=0, last_step=1000, engine=2)
            t0 = time.perf_counter()
            ctx.run_scenario(capi.NB_SCN_MIN_DIST, planet, asteroid, first_step=0, last_step=100000, engine=2)
            k3 = f"; K3 persistent engine {(time.perf_counter() - t0) / 100000 * 1e6:.3f} us/step wall"
    print(f"{case}: n={n} K2 {ms * 1e3:.2f} us/step (5000 back-to-back launches, HIP events); "
          f"per-step-launch engine {wall / 20000 * 1e6:.2f} us/step wall{k3}")

n = 1 << 20
q, v, m = synthetic.bodies(n)
with capi.Context(n, capi.NB_F32, 0, G=synthetic.G, eps=synthetic.EPS, dt=synthetic.DT) as ctx:
    t0 = time.perf_counter(); ctx.set_state(q, v, m); t1 = time.perf_counter()
    ctx.get_state(); t2 = time.perf_counter()
    print(f"N=2^20 F32: nb_set_state {1e3 * (t1 - t0):.1f} ms, nb_get_state {1e3 * (t2 - t1):.1f} ms "
          f"(host conversion + PCIe; one step is ~257 ms)")
