"""Generate the single-step known-answer fixtures from the REFERENCE's own run_step.

Run in the build container only (needs /root/reference, through oracle/_ref/libnbody_ref.so which oracle/Makefile
compiles from /root/reference/samples/nbody.cc where it lies):

    python tests/golden/make_kats.py

Writes tests/golden/kat_<case>.npz holding, for the Problem-2 setting (devices massive, masses time-varying —
samples/nbody.cc:126-130), the full state (q, v as (3,n) float64) after steps 1, 2 and 1000.  The fixtures are
data (inputs are the committed testcases, outputs are the arrays below); no reference source is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = {"b20": (1, 2, 1000), "b1024": (1, 2, 1000), "b200": (1, 2, 1000)}


def main():
    O.build()
    assert O.have_reference(), "oracle/_ref not built: /root/reference missing?"
    for case, steps in CASES.items():
        s = O.read_input(os.path.join(ROOT, "tests/golden/testcases", f"{case}.in"))
        out = {}
        done = 0
        for st in steps:
            O.ref_run_steps(s, done + 1, st - done)
            done = st
            out[f"q_{st}"] = s.q.copy()
            out[f"v_{st}"] = s.v.copy()
        path = os.path.join(ROOT, "tests/golden", f"kat_{case}.npz")
        np.savez_compressed(path, steps=np.array(steps), **out)
        print(case, "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
