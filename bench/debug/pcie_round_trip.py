import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import nbody_amd
from nbody_amd import capi as c, synthetic as syn
n = 1 << 20
q, v, m = syn.bodies(n)
for prec, name in ((c.NB_F32, "NB_F32"), (c.NB_F32_ACC64, "NB_F32_ACC64")):
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
        ctx.set_state(q, v, m); ctx.step(1, 1)
        ts, tg, tk = [], [], []
        for k in range(5):
            t0 = time.perf_counter(); ctx.set_state(q, v, m); t1 = time.perf_counter()
            ms = ctx.step_timed(2 + k, 1); t2 = time.perf_counter()
            ctx.get_state(); t3 = time.perf_counter()
            ts.append(t1 - t0); tk.append(ms); tg.append(t3 - t2)
        print(f"{name}: nb_set_state {1e3*min(ts):.1f} ms  nb_step {min(tk):.1f} ms  nb_get_state {1e3*min(tg):.1f} ms  -> "
              f"{n*(n-1)/(min(ts)+min(tk)*1e-3+min(tg)):.3e} pairs/s with the state crossing PCIe every step, {n*(n-1)/(min(tk)*1e-3):.3e} resident")
