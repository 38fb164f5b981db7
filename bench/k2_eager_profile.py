#!/usr/bin/env python3
"""Kernel-only durations of the per-step engine for rocprofv3 (which, at version 1.1, crashes inside hipGraphLaunch — also on
bench/ubench/launch_rate — so the graph replays themselves cannot be traced): the same kernels issued eagerly
(NB_SCN_EAGER), 4000 steps each of P1-, P2- and Problem-3-type scenarios.
    rocprofv3 --kernel-trace --stats -- python3 bench/k2_eager_profile.py b200 b1024"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c  # noqa: E402
from oracle import oracle as O  # noqa: E402  (input parsing only)

for case in sys.argv[1:] or ["b200", "b1024"]:
    s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", case + ".in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    for kind, watch in ((c.NB_SCN_MIN_DIST, []), (c.NB_SCN_FIRST_HIT, devs), (c.NB_SCN_MISSILE, devs[:1])):
        with c.Context(s.n) as x:
            x.set_state(s.q, s.v, s.m, s.is_device)
            x.run_scenario(kind, s.planet, s.asteroid, last_step=4000, watch=watch, engine=1, flags=c.NB_SCN_EAGER)
    print(case, "done", flush=True)
