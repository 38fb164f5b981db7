#!/usr/bin/env python3
"""bench/small_n_sym_ab.py LIB — K1 (every ordered pair) against K1s (every unordered pair once) for systems BELOW K1s' threshold of
49152 bodies, with a library built with a lower threshold (make LIB=bench/ab/symmin/libnbody_amd.so EXTRA=-DNB_SYM_MIN_N=8192 lib):
the fused step through the raw launch, source_path 2 vs 3, alternating.  Where does K1s start to win?"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

with c.use_library(sys.argv[1]):
    for n in (8192, 12288, 16384, 20480, 24576, 28672, 32768, 36864, 40960, 45056, 49152, 57344, 65536):
        pos, vel = syn.body4_f32(n)
        src = torch.from_numpy(pos).cuda()
        out = torch.zeros_like(src)
        v = torch.from_numpy(vel).cuda()
        stream = torch.cuda.current_stream().cuda_stream
        res = {}
        for sp in (2, 3):
            need = c.workspace_bytes_sym_f32(n) if sp == 3 else c.workspace_bytes_f32(n) * 4
            if need <= 0:
                res[sp] = float("nan")
                continue
            ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            step = lambda: c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, syn.DT, stream, vel_ptr=v.data_ptr(),  # noqa: E731
                                        source_path=sp, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
            best = 1e9
            for rnd in range(3):
                step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = max(10, int(4e10 / (n * n)))
                e0.record()
                for _ in range(reps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            res[sp] = best
        f = lambda ms: 20 * n * (n - 1) / (ms * 1e-3) / 157.3e12  # noqa: E731
        print(f"n = {n:6d} ({-(-n // 4096):2d} superblocks)  K1 {res[2]:8.4f} ms = {f(res[2]):.3f} of peak   K1s {res[3]:8.4f} ms = {f(res[3]):.3f}   "
              f"{'K1s' if res[3] < res[2] else 'K1 '} wins by {abs(res[2] / res[3] - 1) * 100:.1f} %", flush=True)
