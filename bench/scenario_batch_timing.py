#!/usr/bin/env python3
"""Per-step cost of the scenario engines as a function of how many scenarios share a launch stream (run on the GPU box):
K2 batched (one launch per step, blockIdx.y = scenario) and K3 batched (one persistent launch, a workgroup per scenario).
Feeds the scheduling choices in nb_solve: python bench/scenario_batch_timing.py > profiles/r02_scenario_batch_timing.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c  # noqa: E402
from oracle import oracle as O  # noqa: E402  (input parsing only)

STEPS = 20000
for case in sys.argv[1:] or ["b100", "b200", "b512", "b1024"]:
    s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", case + ".in"))
    for engine in ((2, 1) if s.n <= 128 else (1,)):
        line = []
        for k in (1, 2, 3, 4, 5, 6):
            ctxs = [c.Context(s.n) for _ in range(k)]
            for x in ctxs:
                x.set_state(s.q, s.v, s.m, s.is_device)
            kws = [dict(kind=c.NB_SCN_MIN_DIST, planet=s.planet, asteroid=s.asteroid, last_step=STEPS, engine=engine)] * k
            c.run_scenarios_batched(ctxs, [dict(kw, last_step=200) for kw in kws])  # warm-up (code object, tables)
            for x in ctxs:
                x.set_state(s.q, s.v, s.m, s.is_device)
            t0 = time.perf_counter()
            c.run_scenarios_batched(ctxs, kws)
            dt = time.perf_counter() - t0
            line.append("k=%d %.2f" % (k, dt / STEPS * 1e6))
            for x in ctxs:
                x.close()
        print("%s n=%d engine=%s us/step: %s" % (case, s.n, "K3 persistent" if engine == 2 else "K2 per-step", "  ".join(line)),
              flush=True)
