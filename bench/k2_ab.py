#!/usr/bin/env python3
"""Same-device A/B of two builds of the library on the per-step fp64 engine: one scenario (P1) replayed from a hipGraph,
us per step, alternating builds.   python bench/k2_ab.py <other_lib.so> [cases...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c  # noqa: E402
from oracle import oracle as O  # noqa: E402  (input parsing only)

other = os.path.abspath(sys.argv[1])
STEPS = 40000
for case in sys.argv[2:] or ["b200", "b512", "b1024"]:
    s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", case + ".in"))
    res = {"in-tree": [], "other": []}
    for rep in range(3):
        for name, path in (("in-tree", c.library_path()), ("other", other)):
            with c.use_library(path):
                with c.Context(s.n) as x:
                    x.set_state(s.q, s.v, s.m, s.is_device)
                    x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=4000, engine=1)  # warm-up + capture
                    t0 = time.perf_counter()
                    r = x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, first_step=4000, last_step=4000 + STEPS,
                                       engine=1)
                    res[name].append((time.perf_counter() - t0) / STEPS * 1e6)
                    q, _ = x.get_state()
            res.setdefault("q_" + name, q)
    same = (res["q_in-tree"] == res["q_other"]).all()
    print(f"{case} n={s.n}: in-tree {['%.3f' % v for v in res['in-tree']]} us/step, other {['%.3f' % v for v in res['other']]} us/step, "
          f"final positions bit-identical: {bool(same)}", flush=True)
