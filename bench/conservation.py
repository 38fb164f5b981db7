#!/usr/bin/env python3
"""What NB_F32_ACC64 buys (BASELINE configs[4]'s arithmetic): trajectory error against the all-fp64 run (K1-f64 on the GPU)
after `steps` fused steps, for fp32 state + fp32 pair math (NB_F32) and for fp32 pair math with fp64 accumulation and
fp64 q,v masters (NB_F32_ACC64).  All three start from the same fp32-rounded bodies.
    python bench/conservation.py [N=65536] [steps=400] [dt=1e-3]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402
from nbody_amd import capi, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
q, v, m = synthetic.bodies(n)
q = q.astype(np.float32).astype(np.float64)
v = v.astype(np.float32).astype(np.float64)
m = (synthetic.G * m).astype(np.float32).astype(np.float64) / synthetic.G  # the fp32 modes fold G*m once, rounded
out = {}
for name, prec in (("NB_F64", capi.NB_F64), ("NB_F32", capi.NB_F32), ("NB_F32_ACC64", capi.NB_F32_ACC64)):
    with capi.Context(n, prec, 0, G=synthetic.G, eps=synthetic.EPS, dt=dt, f64_large_min=1024) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, steps)
        out[name] = ctx.get_state()
q64, v64 = out["NB_F64"]
move = np.sqrt(((q64 - q) ** 2).sum(axis=0)).mean()
print(f"N={n} steps={steps} dt={dt}: mean displacement over the run {move:.3e} (positions O(1), fp32 ulp 6e-8)")
for name in ("NB_F32", "NB_F32_ACC64"):
    qq, vv = out[name]
    eq = np.sqrt(((qq - q64) ** 2).sum(axis=0))
    ev = np.sqrt(((vv - v64) ** 2).sum(axis=0))
    print(f"{name:13s} vs fp64: position error rms {np.sqrt((eq ** 2).mean()):.3e} max {eq.max():.3e} "
          f"({np.sqrt((eq ** 2).mean()) / move:.2e} of the displacement); velocity error rms {np.sqrt((ev ** 2).mean()):.3e}")
