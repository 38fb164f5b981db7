#!/usr/bin/env python3
"""bench/fuzz_sharded.py [cases=30] [seed=3] — one-off randomized sweep of the native multi-GPU host (nb_sharded_*) with all ranks on
ONE GPU: random rank counts (1-8), body counts (whole superblocks per shard or not: shared pairs or ordered pairs), exchanges
(copy / host-staged), precisions, plain or overlapped, with and without a deadline — two steps against nb_step of an unsharded
context (same arithmetic, sums cut elsewhere: fp32 rounding).  Prints a line per case and a verdict."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

SB = 4096


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    bad = 0
    for k in range(cases):
        P = int(rng.integers(1, 9))
        acc64 = bool(rng.integers(2))
        if rng.integers(2):  # whole superblocks per shard, enough bodies: the ranks share the unordered pairs
            lo = max(1, -(-12 // P))
            n = P * SB * int(rng.integers(lo, max(lo + 1, 10)))
        else:                # anything divisible by P (and by 256 P when overlapped)
            n = P * 256 * int(rng.integers(2, 120))
        exchange = str(rng.choice(["copy", "copy", "host"]))
        overlap = bool(rng.integers(2)) and exchange == "copy" and P > 1
        deadline = float(rng.choice([0.0, 60.0]))
        prec = c.NB_F32_ACC64 if acc64 else c.NB_F32
        q, v, m = syn.bodies(n)
        with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=1e-2) as ctx:
            ctx.set_state(q, v, m)
            ctx.step(1, 2)
            q_ref, v_ref = ctx.get_state()
        with c.Sharded(n, [0] * P, prec, G=syn.G, eps=syn.EPS, dt=1e-2, overlap=overlap, exchange=exchange, deadline=deadline) as sh:
            name = sh.kernel_name()
            sh.set_state(q, v, m)
            sh.step(2)
            q2, v2 = sh.get_state()
        dq, dv = float(np.abs(q2 - q_ref).max()), float(np.abs(v2 - v_ref).max())
        ok = np.isfinite(q2).all() and dq < (2e-7 if acc64 else 5e-7) and dv < 1e-5
        bad += not ok
        print(f"case {k:3d}  P = {P}  n = {n:7d}  acc64 {int(acc64)}  {exchange:4s} overlap {int(overlap)} deadline {deadline:4.0f}  "
              f"{name.split('<')[0]:22s} max|dq| {dq:.1e} max|dv| {dv:.1e}  {'ok' if ok else 'FAIL'}", flush=True)
    print(f"{cases} cases, {bad} failed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
