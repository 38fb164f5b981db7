#!/usr/bin/env python3
"""bench/fuzz_sym.py [cases=40] [seed=5] — one-off randomized sweep of K1s (every unordered pair once) against K1 (every ordered
pair) on ALL bodies, through the C ABI on one GPU: random body counts (ragged last superblock, odd / even superblock counts),
forced workgroup counts per superblock, both precisions; and the multi-GPU form — random rank counts, every rank's partial force
summed — against the same K1 result.  The suite pins a handful of shapes against the oracle (tests/test_gpu_f32_symmetric.py);
this looks for a shape that breaks the pair schedule or the ragged-end handling.  Prints one line per case and a verdict."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

SB = 4096


def accel(src, n, acc64, source_path, j_split=0):
    need = c.workspace_bytes_sym_f32(n, acc64) if source_path == 3 else c.workspace_bytes_f32(n, acc64)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    a = torch.zeros((n, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
    c.launch_f32(src.data_ptr(), 0, n, 0, n, syn.EPS ** 2, syn.DT, torch.cuda.current_stream().cuda_stream, accel_only=True,
                 acc_ptr=a.data_ptr(), acc64=acc64, source_path=source_path, j_split=j_split, workspace_ptr=ws.data_ptr(),
                 workspace_bytes=ws.numel())
    torch.cuda.synchronize()
    return a[:, :3].double()


def shares(src, n, ranks, acc64):
    per = n // ranks
    ws = torch.empty(c.workspace_bytes_shared_pairs_f32(n, ranks, acc64), dtype=torch.uint8, device="cuda")
    part = torch.empty((n, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
    total = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    for r in range(ranks):
        part.fill_(float("nan"))
        c.launch_pair_forces_f32(src.data_ptr(), n, r * per, per, syn.EPS ** 2, torch.cuda.current_stream().cuda_stream,
                                 part.data_ptr(), ws.data_ptr(), ws.numel(), acc64=acc64)
        total += part[:, :3].double()
    torch.cuda.synchronize()
    return total


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    worst, bad = 0.0, 0
    for k in range(cases):
        acc64 = bool(rng.integers(2))
        multi = k % 3 == 2
        if multi:
            ranks = int(rng.integers(2, 9))
            n = ranks * int(rng.integers(max(2, -(-9 // ranks)), 12)) * SB
            while n * n < 1.1e9 * ranks:  # (below that the ranks do not share the pairs: include/nbody_amd_ext.h)
                n += ranks * SB
            what = f"{ranks} ranks"
        else:
            n = int(rng.integers(7 * SB, 80 * SB)) - int(rng.integers(0, SB)) * int(rng.integers(2))
            n = max(n, 28672)
            chunks = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
            what = f"chunks {chunks}"
        pos, _ = syn.body4_f32(n)
        src = torch.from_numpy(pos).cuda()
        ref = accel(src, n, acc64, 2)
        got = shares(src, n, ranks, acc64) if multi else accel(src, n, acc64, 3, chunks)
        scale = ref.abs().max().item()
        err = ((got - ref).abs().max() / scale).item() if torch.isfinite(got).all() else float("inf")
        ok = err < (2e-6 if not acc64 else 2e-6)  # two kernels, two summation orders, fp32 pair arithmetic in both
        worst, bad = max(worst, err if err != float("inf") else 0.0), bad + (not ok)
        print(f"case {k:3d}  n = {n:7d} (B = {-(-n // SB):3d}, ragged {(-n) % SB:4d})  acc64 {int(acc64)}  {what:10s}  "
              f"max|K1s - K1| / max|a| = {err:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    print(f"{cases} cases, {bad} failed, worst {worst:.2e}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
