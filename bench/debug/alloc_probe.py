"""bench/debug/alloc_probe.py — create / set_state / workspace / first step / second step / destroy of an fp32 context of 2^20 bodies, six times in one
process: does the lazily allocated K1s workspace (1.6 GB, hipMalloc right after the previous context freed as much) ever stall the first step?
Round 5: no — 172-175 ms against 170-171 ms for the second step (one 5.7 s first step seen once in bench/create_timing.py was not reproduced)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import nbody_amd
from nbody_amd import capi as c, synthetic as syn
torch.cuda.init()
n = 1 << 20
q, v, m = syn.bodies(n)
for k in range(6):
    t0 = time.perf_counter()
    ctx = c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=syn.DT)
    t1 = time.perf_counter()
    ctx.set_state(q, v, m)
    t2 = time.perf_counter()
    name = ctx.kernel_name()          # allocates the workspace
    t3 = time.perf_counter()
    ctx.step(1, 1)
    t4 = time.perf_counter()
    ctx.step(2, 1)
    t5 = time.perf_counter()
    ctx.close()
    t6 = time.perf_counter()
    print(f"round {k}: create {1e3*(t1-t0):7.1f}  set_state {1e3*(t2-t1):7.1f}  workspace {1e3*(t3-t2):8.1f}  step1 {1e3*(t4-t3):8.1f}  step2 {1e3*(t5-t4):7.1f}  destroy {1e3*(t6-t5):7.1f} ms", flush=True)
