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

STEPS = 40000
for case in sys.argv[1:] or ["b100", "b200", "b512", "b1024"]:
    s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", case + ".in"))
    for engine, flags in (((2, 0), (1, 0), (1, c.NB_SCN_EAGER)) if s.n <= 128 else ((1, 0), (1, c.NB_SCN_EAGER))):
        with c.Context(s.n) as x:  # the unbatched driver (nb_run_scenario) for reference
            x.set_state(s.q, s.v, s.m, s.is_device)
            x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=200, engine=engine, flags=flags)
            x.set_state(s.q, s.v, s.m, s.is_device)
            t0 = time.perf_counter()
            x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=STEPS, engine=engine, flags=flags)
            single = (time.perf_counter() - t0) / STEPS * 1e6
        line = ["single %.2f" % single]
        for k in (1, 2, 3, 4, 5, 6):
            ctxs = [c.Context(s.n) for _ in range(k)]
            for x in ctxs:
                x.set_state(s.q, s.v, s.m, s.is_device)
            kws = [dict(kind=c.NB_SCN_MIN_DIST, planet=s.planet, asteroid=s.asteroid, last_step=STEPS, engine=engine,
                        flags=flags)] * k
            c.run_scenarios_batched(ctxs, [dict(kw, last_step=200) for kw in kws])  # warm-up (code object, tables)
            for x in ctxs:
                x.set_state(s.q, s.v, s.m, s.is_device)
            t0 = time.perf_counter()
            c.run_scenarios_batched(ctxs, kws)
            dt = time.perf_counter() - t0
            line.append("k=%d %.2f" % (k, dt / STEPS * 1e6))
            for x in ctxs:
                x.close()
        name = "K3 persistent" if engine == 2 else "K2 eager launches" if flags else "K2 graph replay"
        print("%s n=%d engine=%s us/step (incl. graph capture): %s" % (case, s.n, name, "  ".join(line)),
              flush=True)
