#!/usr/bin/env python3
"""bench/k1_small_n_model.py — K1 (every ordered pair) below K1s' threshold: the fused step with EVERY source-slice count 1..64 forced,
by system size, against a two-parameter model of the launch: a 256-thread workgroup of 1024 targets that meets t source tiles costs
c0 + t (in units of a tile; c0 = prologue + epilogue), and a CU that carries k such workgroups at once runs each g(k) = 1 + s (k - 1)
times slower.  Prints, per size: the plan's choice, the measured best, what the model would pick and how far each is from the best;
at the end the (c0, s) that minimise the model's worst regret over all sizes."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nbody_amd  # noqa: E402,F401
from nbody_amd import capi as c, synthetic as syn  # noqa: E402

CUS = torch.cuda.get_device_properties(0).multi_processor_count
RANKS = 1  # --ranks P: the step of ONE rank of P (targets = the first n / P bodies, sources = all n): does the model hold for shards?
if len(sys.argv) > 2 and sys.argv[1] == "--ranks":
    RANKS = int(sys.argv[2])
    del sys.argv[1:3]
SIZES = [int(a) for a in sys.argv[1:]] or list(range(4096, 36864 + 1, 2048))


def model_cost(blocks, ntiles, js, c0, s):
    t = -(-ntiles // js)
    eff = -(-ntiles // t)
    k = -(-blocks * eff // CUS)
    return (1 + s * (k - 1)) * (c0 + t)


table = {}
for n in SIZES:
    pos, vel = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    out = torch.zeros_like(src)
    v = torch.from_numpy(vel).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    per = n // RANKS
    v = v[:per].contiguous()
    ws = torch.empty(66 * per * 16, dtype=torch.uint8, device="cuda")  # room for 64 slices in one launch
    auto = c.plan_f32(n, per, False, 0, 0, ws.numel(), 2)
    blocks, ntiles = -(-per // 1024), -(-n // 256)
    res = {}
    seen = {}
    for js in [0] + list(range(1, min(64, ntiles) + 1)):
        if js:  # slice counts that cut the tiles the same way are the same launch up to empty workgroups: measure one of them
            t = -(-ntiles // js)
            if t in seen:
                continue
            seen[t] = js
        step = lambda: c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, per, syn.EPS ** 2, syn.DT, stream, vel_ptr=v.data_ptr(),  # noqa: E731
                                    source_path=2, j_split=js, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
        best = 1e9
        for rnd in range(3):
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                step()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        res[js] = best
    table[n] = (blocks, ntiles, auto[1], res)
    b = min((k for k in res if k), key=res.get)
    print(f"n = {n:6d}{'' if RANKS == 1 else ' / %d ranks' % RANKS}  {blocks:2d} blocks {ntiles:3d} tiles  plan {auto[1]:2d} slices {res[0]:.4f} ms  best {b:2d} slices {res[b]:.4f} ms "
          f"(plan {100 * (res[0] / res[b] - 1):+.0f} %)   " + " ".join(f"{k}:{t:.4f}" for k, t in res.items() if k), flush=True)


def regret(c0, s):
    worst, rows = 0.0, []
    for n, (blocks, ntiles, _, res) in table.items():
        cands = [k for k in res if k]
        pick = min(cands, key=lambda js: (model_cost(blocks, ntiles, js, c0, s), js))
        r = res[pick] / min(res[k] for k in cands) - 1
        rows.append((n, pick, r))
        worst = max(worst, r)
    return worst, rows


fits = sorted((regret(c0 / 100, s / 100)[0], c0 / 100, s / 100) for c0 in range(0, 301, 10) for s in range(30, 101, 5))
print("\nmodel fits (worst regret over the sizes, c0, s):", [(round(w, 3), a, b) for w, a, b in fits[:8]])
for c0, s in ((0.66, 0.70), (fits[0][1], fits[0][2])):
    w, rows = regret(c0, s)
    print(f"c0 = {c0}, s = {s}: worst {100 * w:.1f} %  " + " ".join(f"{n}:{p}({100 * r:+.0f}%)" for n, p, r in rows))
