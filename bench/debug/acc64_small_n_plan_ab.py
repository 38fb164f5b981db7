"""bench/debug/acc64_small_n_plan_ab.py — contexts of small whole systems in both fp32 modes, the product library (K1 slices from the co-residency model)
against a build with rounds 2-5's rule (make LIB=bench/ab/oldslices/libnbody_amd.so EXTRA=-DNB_K1_SLICE_MODEL=0 lib): the model was fitted on NB_F32;
does NB_F32_ACC64 (double4 partial records, a heavier reducer) gain the same?"""
import sys
sys.path.insert(0, '/root/repo')
import nbody_amd
from nbody_amd import capi as c, synthetic as syn
def t(n, prec):
    q, v, m = syn.bodies(n)
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 20)
        return min(ctx.step_timed(21 + 200 * r, 200) for r in range(3)), ctx.kernel_name()
for n in (2048, 6144, 8192, 10240, 22528, 24576, 26624):
    row = f"n = {n:6d}"
    for lib in (None, 'bench/ab/oldslices/libnbody_amd.so'):
        cm = c.use_library(lib) if lib else __import__('contextlib').nullcontext()
        with cm:
            for prec, nm in ((c.NB_F32, 'f32'), (c.NB_F32_ACC64, 'acc64')):
                ms, k = t(n, prec)
                row += f"  | {'old' if lib else 'new'} {nm} {ms:.4f} ms"
    print(row, flush=True)
