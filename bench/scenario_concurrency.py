#!/usr/bin/env python3
"""Two (or more) independent scenario streams at once: per-step cost of each when they share the chip, and when each is
confined to half of the compute units (nb_create_cu_masked, include/nbody_amd_debug.h: the instrumented build).  Feeds nb_solve's stream layout for
n > 256.   python bench/scenario_concurrency.py [b1024 ...]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c  # noqa: E402
from oracle import oracle as O  # noqa: E402  (input parsing only)

c.use_library(c.stamps_library_path()).__enter__()  # the masks are a hook of the instrumented build
STEPS = 40000


def run(case, masks):
    s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", case + ".in"))
    ctxs = []
    how = {None: c.NB_CU_ALL, "lo": c.NB_CU_LOW, "hi": c.NB_CU_HIGH, "even": c.NB_CU_EVEN, "odd": c.NB_CU_ODD}
    for m in masks:
        x = c.Context(s.n, cu_mask=how[m])
        x.set_state(s.q, s.v, s.m, s.is_device)
        ctxs.append(x)
    out = [0.0] * len(ctxs)

    def work(k):
        t0 = time.perf_counter()
        ctxs[k].run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=STEPS, engine=1)
        out[k] = (time.perf_counter() - t0) / STEPS * 1e6

    for x in ctxs:  # warm-up: graph capture etc. happen per run, keep them out of the comparison as far as possible
        x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=200, engine=1)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(ctxs))]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = (time.perf_counter() - t0) / STEPS * 1e6
    for x in ctxs:
        x.close()
    return out, wall


for case in sys.argv[1:] or ["b512", "b1024"]:
    for masks in ([None], [None, None], ["lo", "hi"], ["even", "odd"], [None, None, None], ["lo", "hi", "hi"]):
        per, wall = run(case, masks)
        print("%s streams=%s us/step each: %s   (all done after %.2f us/step)" %
              (case, [m or "all" for m in masks], "  ".join("%.2f" % p for p in per), wall), flush=True)
