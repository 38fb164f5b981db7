"""K1s — the fp32 force that evaluates every UNORDERED pair once (Newton's third law; csrc/nbody_kernels_f32_sym.hip) —
against the fp64 oracle and against K1 (every ordered pair), through the C ABI.

The reference sums a_i over every j != i (samples/nbody.cc:57-73); K1s produces the same sums from half the pair
evaluations: the reaction on the source is accumulated in registers that travel through the wave.  Same tolerances as K1
(tests/test_gpu_f32_parity.py): |a_gpu - a_ref|_inf <= 1e-5 * sum_j |a_ij| (fp32), 1e-6 (fp32 pair math, fp64 accumulation).
"""
import numpy as np
import pytest

from test_gpu_f32_parity import TOL_ACC64, TOL_F32, _check_step_rows, _oracle_rows

pytestmark = pytest.mark.gpu

SB = 4096  # superblock of the kernel


def _launch(nb, torch, src, n, acc64, source_path, accel=None, j_split=0, out=None, vel=None, pos64=None, vel64=None, dt=None):
    c, syn = nb.capi, nb.synthetic
    need = c.workspace_bytes_sym_f32(n, acc64) if source_path == 3 else c.workspace_bytes_f32(n, acc64)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    kw = dict(acc64=acc64, source_path=source_path, j_split=j_split, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
    stream = torch.cuda.current_stream().cuda_stream
    if accel is not None:
        c.launch_f32(src.data_ptr(), 0, n, 0, n, syn.EPS ** 2, syn.DT, stream, accel_only=True, acc_ptr=accel.data_ptr(), **kw)
    else:
        c.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, dt, stream,
                     vel_ptr=vel.data_ptr() if vel is not None else 0, pos64_ptr=pos64.data_ptr() if pos64 is not None else 0,
                     vel64_ptr=vel64.data_ptr() if vel64 is not None else 0, **kw)
    torch.cuda.synchronize()
    return ws


@pytest.mark.parametrize("n,chunks", [(7 * SB, 0),           # the smallest system K1s takes (end of round 5: 28672 bodies; B = 7, 32 workgroups each)
                                      (8 * SB - 1234, 0),    # B = 8, ragged: 4-phase pieces, several cuts inside one round
                                      (9 * SB, 0),           # (the threshold of mid-round 5: 36864 bodies; B = 9, odd)
                                      (10 * SB - 100, 0),    # B = 10, even, ragged
                                      (12 * SB, 0),          # rounds 3-4's smallest: 7 work units per superblock cut into
                                                             # sub-unit chunks (>= 4 of a unit's 32 tile phases each; 8 until round 5)
                                      (16 * SB + 3, 0),      # B = 17, odd, ragged
                                      (64 * SB, 0),          # B = 64: even, the half round B/2 for b < 32
                                      (65 * SB + 77, 0),     # B = 66 with a ragged last superblock (4019 bodies missing)
                                      (67 * SB, 3),          # B = 67: odd, no half round; three workgroups per superblock
                                      (64 * SB + 1, 1)])     # B = 65: one body in the last superblock; one workgroup each
@pytest.mark.parametrize("acc64", [False, True])
def test_symmetric_accelerations_vs_oracle_and_k1(nb, oracle, n, chunks, acc64):
    import torch
    c, syn = nb.capi, nb.synthetic
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    dt_acc = torch.float64 if acc64 else torch.float32
    a_sym = torch.zeros((n, 4), dtype=dt_acc, device="cuda")
    a_k1 = torch.zeros((n, 4), dtype=dt_acc, device="cuda")
    ws_bytes = c.workspace_bytes_sym_f32(n, acc64)
    assert ws_bytes >= ((n + SB - 1) // SB // 2 + 1) * n * 12
    assert c.kernel_name_f32(n, n, acc64, workspace_bytes=ws_bytes, accel_only=True, source_path=3) == \
        f"nbody_force_sym_f32<{'true' if acc64 else 'false'}>"
    tpl, js, wg = c.plan_f32(n, n, acc64, 0, chunks, ws_bytes, 3)
    assert (tpl, wg) == (8, 512) and (js == chunks if chunks else js >= 1)
    _launch(nb, torch, src, n, acc64, 3, accel=a_sym, j_split=chunks)
    _launch(nb, torch, src, n, acc64, 2, accel=a_k1)
    a1, a2 = a_sym.cpu().numpy()[:, :3].astype(np.float64), a_k1.cpu().numpy()[:, :3].astype(np.float64)
    assert np.isfinite(a1).all()
    # rows of the first, a middle and the last superblock (its ragged end included), and of both halves of the round list
    B = (n + SB - 1) // SB
    rows = np.unique(np.clip(np.concatenate([np.arange(4), [SB - 1, SB, (B // 2) * SB - 1, (B // 2) * SB + 5,
                                                              (B - 1) * SB - 1, (B - 1) * SB], n - 1 - np.arange(4),
                                             np.arange(24) * (n // 24) + 11]), 0, n - 1))
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    tol = TOL_ACC64 if acc64 else TOL_F32
    err = (np.abs(a1[rows].T - ref).max(axis=0) / s).max()
    assert err < tol, err
    # what it actually delivers, with room (measured 3e-8 / 1e-8 at 2^20; the 1 ulp of v_rsq_f32 does not average out over the
    # few thousand sources of the smallest systems: 2.4e-7 at 40860 bodies with fp64 sums)
    assert err < (5e-7 if not acc64 else (4e-7 if n < 12 * SB else 2e-7)), err
    # every body against K1 (every ordered pair): two summation orders of the same terms
    scale = np.abs(a2).max()
    assert np.abs(a1 - a2).max() < 2e-5 * scale, np.abs(a1 - a2).max() / scale
    # Newton's third law holds pair by pair here: the total momentum rate cancels to rounding
    gm = pos[:, 3].astype(np.float64)
    p = (a1 * gm[:, None]).sum(axis=0)
    assert np.all(np.abs(p) < 1e-5 * (np.abs(a1) * gm[:, None]).sum(axis=0)), p


def test_symmetric_is_bitwise_reproducible_and_is_what_a_context_runs(nb, oracle):
    """No atomics anywhere: two launches give identical bits; nb_accel / nb_step of a context with >= 28672 bodies pick
    K1s by themselves (the workspace is sized for it at nb_create) and give exactly the raw launch's numbers."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n = 70 * SB + 5
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    a = [torch.zeros((n, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    for k in range(2):
        _launch(nb, torch, src, n, False, 3, accel=a[k])
    assert torch.equal(a[0], a[1])
    q, v, m = syn.bodies(n)
    with c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
        ctx.set_state(q, v, m)
        a_ctx = ctx.accel(1)
    assert np.array_equal(a_ctx.T.astype(np.float32), a[0].cpu().numpy()[:, :3])


@pytest.mark.parametrize("acc64", [False, True])
def test_symmetric_step_against_oracle(nb, oracle, acc64):
    """One fused step (force, reducer, kick-drift epilogue): v' = v + a*dt, q' = q + v'*dt (samples/nbody.cc:76-88) on rows
    of a ragged system, and nb_step of a context equals the raw launch bit for bit."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n, dt = 66 * SB + 1234, 1e-2
    pos, vel_np = syn.body4_f32(n)
    q, v, m = syn.bodies(n)
    src = torch.from_numpy(pos).cuda()
    out = torch.zeros_like(src)
    rows = np.concatenate([np.arange(30) * (n // 30) + 7, [0, SB - 1, SB, n - 1235, n - 1]])
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    if not acc64:
        vel = torch.from_numpy(vel_np).cuda()
        _launch(nb, torch, src, n, False, 3, out=out, vel=vel, dt=dt)
        _check_step_rows(pos, vel_np, out.cpu().numpy(), vel.cpu().numpy(), rows, 0, ref, s, dt, TOL_F32)
    else:
        p64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([q.T, (syn.G * m)[:, None]], axis=1))).cuda()
        v64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([v.T, np.zeros((n, 1))], axis=1))).cuda()
        _launch(nb, torch, src, n, True, 3, out=out, pos64=p64, vel64=v64, dt=dt)
        dt32 = np.float64(np.float32(dt))
        v_new = v64.cpu().numpy()[rows, :3]
        a_gpu = (v_new - v.T[rows]) / dt32
        assert (np.abs(a_gpu.T - ref).max(axis=0) / s).max() < TOL_ACC64
        assert np.all(np.abs(p64.cpu().numpy()[rows, :3] - (q.T[rows] + v_new * dt32)) <= 4e-16)
        assert np.array_equal(out.cpu().numpy()[rows, :3], p64.cpu().numpy()[rows, :3].astype(np.float32))
    with c.Context(n, c.NB_F32_ACC64 if acc64 else c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=dt) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 1)
        qg, vg = ctx.get_state()
    if acc64:
        assert np.array_equal(qg.T, p64.cpu().numpy()[:, :3]) and np.array_equal(vg.T, v64.cpu().numpy()[:, :3])
    else:
        assert np.array_equal(qg.T.astype(np.float32), out.cpu().numpy()[:, :3])
        assert np.array_equal(vg.T.astype(np.float32), vel.cpu().numpy()[:, :3])


def test_symmetric_refusals_and_fallbacks(nb):
    """source_path 3 is refused where K1s cannot run (a target window, too few bodies, too small a workspace, a phase);
    auto silently takes K1 there."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n = 64 * SB
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    acc = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    big = torch.empty(c.workspace_bytes_sym_f32(n), dtype=torch.uint8, device="cuda")
    small = torch.empty(c.workspace_bytes_f32(n), dtype=torch.uint8, device="cuda")
    assert 9e9 < c.workspace_bytes_sym_f32(1 << 24) < 13e9  # 412 GB of slots in one launch: 128 batches of 32 superblocks within 720 B per body
    assert c.workspace_bytes_sym_f32(SB * 7 - 1) == 0 and c.workspace_bytes_sym_f32(SB * 7) > 0 and c.workspace_bytes_sym_f32(1 << 28) == 0  # too small / no batch of superblocks fits 128 GiB
    for kw in (dict(workspace_ptr=small.data_ptr(), workspace_bytes=small.numel()),                      # workspace too small
               dict(workspace_ptr=big.data_ptr(), workspace_bytes=big.numel(), src_begin=0, src_end=n // 2),  # a source range
               dict(workspace_ptr=big.data_ptr(), workspace_bytes=big.numel(), tgt_ptr=src.data_ptr())):  # a target block
        with pytest.raises(c.NBodyError, match="unordered pair") as e:
            c.launch_f32(src.data_ptr(), 0, n, 0, n, syn.EPS ** 2, syn.DT, stream, accel_only=True, acc_ptr=acc.data_ptr(),
                         source_path=3, **kw)
        assert e.value.code == c.NB_ERR_INVALID
    with pytest.raises(c.NBodyError):  # a window of the targets
        c.launch_f32(src.data_ptr(), 0, n, 4096, n - 4096, syn.EPS ** 2, syn.DT, stream, accel_only=True,
                     acc_ptr=acc.data_ptr(), source_path=3, workspace_ptr=big.data_ptr(), workspace_bytes=big.numel())
    # auto: the small workspace gives K1, the big one K1s; a forced register blocking asks for K1
    assert c.kernel_name_f32(n, n, workspace_bytes=small.numel()).startswith("nbody_force_f32<")
    assert c.kernel_name_f32(n, n, workspace_bytes=big.numel()) == "nbody_force_sym_f32<false>"
    assert c.kernel_name_f32(n, n, targets_per_lane=8, workspace_bytes=big.numel()).startswith("nbody_force_f32<")
    assert c.kernel_name_f32(n, n, j_split=8, workspace_bytes=big.numel()).startswith("nbody_force_f32<")
    assert c.kernel_name_f32(n, n, workspace_bytes=big.numel(), source_path=2).startswith("nbody_force_f32<")


def test_batched_launches_of_a_system_too_large_for_a_slot_per_round(nb, oracle):
    """N = 2^23 on one GPU: a slot per round would be 103 GB.  Beyond 2 GiB of slots the I-superblocks go in batches — each a launch
    like one rank of a multi-GPU step, whose reducer adds the batch's slots to a running force behind them — and the last batch's
    reducer runs the epilogue.  Round 5: the batches are sized for 720 B per body (64 batches of 32 superblocks x 8 workgroups, 5 GB;
    rounds 3-4: 4 x 512 in 52 GB, 0.8 % slower).  nb_accel of a context against 8 oracle rows from the first, middle and last
    batches, and against the momentum identity."""
    c, syn = nb.capi, nb.synthetic
    n = 1 << 23
    assert 4e9 < c.workspace_bytes_sym_f32(n) < 7e9
    c.selftest_pair_schedule(n, 256, 1)
    q, v, m = syn.bodies(n)
    with c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
        ctx.set_state(q, v, m)
        a = ctx.accel(1)
    assert np.isfinite(a).all()
    q32 = np.ascontiguousarray(q.astype(np.float32).astype(np.float64))
    gm = (syn.G * m).astype(np.float32).astype(np.float64) / syn.G
    rows = [0, 4095, 512 * SB - 1, 512 * SB, n // 2 + 77, 3 * 512 * SB + 5, n - SB, n - 1]
    ref, s = oracle.accel_rows_at(q32, gm, syn.G, syn.EPS, rows, want_abs=True)
    worst = float((np.abs(a[:, rows] - ref).max(axis=0) / s).max())
    assert worst < 2e-7, worst  # (tolerance 1e-5; K1s delivers 3e-8)
    p = (a * gm).sum(axis=1)
    assert np.all(np.abs(p) < 1e-5 * (np.abs(a) * gm).sum(axis=1)), p


def test_symmetric_on_a_clustered_system_with_a_wide_mass_range(nb, oracle):
    """Not the uniform cloud of the bench: bodies concentrated towards the centre (r -> r^3: a 10^4-fold density contrast,
    thousands of pairs inside the softening length) with masses spread over three decades.  The reaction sums of K1s pass
    through fp32 accumulators that travel and an fp32 LDS image; the tolerance of SURVEY 8(d) must hold here too, and K1s must
    agree with K1."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n = 16 * SB
    rng = np.random.default_rng(7)
    q = rng.uniform(-1, 1, size=(n, 3))
    q *= (np.linalg.norm(q, axis=1, keepdims=True) ** 2)
    gm = 10.0 ** rng.uniform(-3, 0, size=n) / n
    pos = np.ascontiguousarray(np.concatenate([q, gm[:, None]], axis=1).astype(np.float32))
    src = torch.from_numpy(pos).cuda()
    out = {}
    for acc64 in (False, True):
        a_sym = torch.zeros((n, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
        a_k1 = torch.zeros_like(a_sym)
        _launch(nb, torch, src, n, acc64, 3, accel=a_sym)
        _launch(nb, torch, src, n, acc64, 2, accel=a_k1)
        a1, a2 = a_sym.cpu().numpy()[:, :3].astype(np.float64), a_k1.cpu().numpy()[:, :3].astype(np.float64)
        assert np.isfinite(a1).all()
        r = np.linalg.norm(pos[:, :3], axis=1)
        rows = np.concatenate([np.argsort(r)[:12], np.argsort(r)[-6:], np.arange(12) * (n // 12) + 5])  # the dense core first
        ref, s = _oracle_rows(oracle, syn, pos, rows)
        e_sym = (np.abs(a1[rows].T - ref).max(axis=0) / s).max()
        e_k1 = (np.abs(a2[rows].T - ref).max(axis=0) / s).max()
        out[acc64] = (e_sym, e_k1)
        assert e_sym < (TOL_ACC64 if acc64 else TOL_F32), out
        assert e_sym < 10 * max(e_k1, 3e-8), out  # no worse than the ordered-pair kernel by more than its own noise level
    print("clustered: max err / sum|a_ij|  (K1s, K1)  fp32:", out[False], " fp64-accumulated:", out[True])


# ---------------------------------------------------------------- the default 8-rank shapes of the BASELINE configs, on one GPU

def _eight_rank_shares_vs_oracle(nb, oracle, n, acc64, tol):
    """What eight GPUs compute for ONE step of `n` bodies when they share the unordered pairs (the default of both hosts):
    rank r = 0..7 runs nb_launch_pair_forces_f32 on the superblocks of its shard — here one after the other on the one GPU —
    and the reduce-scatter adds the eight partial forces.  NaN-prefilled outputs prove that every rank's launch writes every
    body; the sum is checked on 24 rows, three from every rank's shard, against the fp64 oracle (samples/nbody.cc:57-73)."""
    import torch
    c, syn = nb.capi, nb.synthetic
    ranks = 8
    per = n // ranks
    c.selftest_pair_schedule(n, 256, ranks, acc64)
    ws_bytes = c.workspace_bytes_shared_pairs_f32(n, ranks, acc64)
    nb_, wg, sub = c.plan_shared_pairs_f32(n, ranks, acc64)
    assert nb_ == per // SB and wg >= 1 and sub >= 1
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    dt_acc = torch.float64 if acc64 else torch.float32
    part = torch.empty((n, 4), dtype=dt_acc, device="cuda")
    total = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(ranks):
        part.fill_(float("nan"))
        c.launch_pair_forces_f32(src.data_ptr(), n, r * per, per, syn.EPS ** 2, stream, part.data_ptr(), ws.data_ptr(),
                                 ws.numel(), acc64=acc64)
        total += part  # (the reduce-scatter sums in the collective's own order; fp64 here so that the check sees the shares)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(total[:, :3]).all()), "a rank's launch left bodies unwritten"
    rows = np.array([r * per + off for r in range(ranks) for off in (0, per // 2 + 37 * r + 1, per - 1)])
    a = total[torch.from_numpy(rows).cuda(), :3].cpu().numpy().T
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    err = (np.abs(a - ref).max(axis=0) / s).max()
    assert err < tol, err
    return (nb_, wg, sub), ws_bytes, err


def test_configs3_eight_rank_shares_against_the_oracle(nb, oracle):
    """BASELINE configs[3]: N = 2^22 over 8 GPUs, fp32.  128 superblocks per rank in 4 sub-launches of 32 (8 workgroups each),
    2.4 GB of slots (rounds 3-4: one launch, 6.6 GB, the same speed)."""
    plan, ws_bytes, err = _eight_rank_shares_vs_oracle(nb, oracle, 1 << 22, False, TOL_F32)
    assert plan == (128, 8, 4) and 2e9 < ws_bytes < 3e9 and err < 2e-7, (plan, ws_bytes, err)


@pytest.mark.heavy
def test_configs4_eight_rank_shares_against_the_oracle(nb, oracle):
    """BASELINE configs[4]: N = 2^24 over 8 GPUs, fp32 pair math / fp64 sums.  512 superblocks per rank = 103 GB of slots in one
    launch, so every rank's share goes out as sub-launches, each adding to the partial force of the one before: 16 of 32
    superblocks x 8 workgroups in 11.3 GB (round 5; round 4: 2 of 256 in 52 GB, the same speed; until round 4 this mechanism was
    only executed at N = 2^23 over 2 ranks; this is the real shape, and that stand-in test is gone).  ~45 s of kernels (an eighth
    of the step each); `-m "gpu and not heavy"` deselects it."""
    plan, ws_bytes, err = _eight_rank_shares_vs_oracle(nb, oracle, 1 << 24, True, TOL_ACC64)
    assert plan == (512, 8, 16) and 9e9 < ws_bytes < 13e9 and err < 2e-7, (plan, ws_bytes, err)


# ---------------------------------------------------------------- the context's K1s workspace: lazy, optional, never fatal

def test_context_workspace_is_lazy_optional_and_never_fatal(nb, oracle):
    """nb_create no longer allocates K1s' pair slots (1.7 GB at N = 2^20, 26 GB at 2^22): the first nb_step / nb_accel does.
    NB_CFG_ORDERED_PAIRS opts out (K1 for the context's life).  A device that cannot spare the slots makes the context fall
    back to K1 — said in nb_last_error, no call fails."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n = 1 << 20
    q, v, m = syn.bodies(n)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # (blocks cached by earlier tests would serve the big allocations below without taking device memory)
    used = lambda: torch.cuda.mem_get_info(0)[1] - torch.cuda.mem_get_info(0)[0]  # noqa: E731  (device-wide, not torch's)
    u0 = used()
    ctx = c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=1e-2)
    ctx.set_state(q, v, m)
    u1 = used()
    assert u1 - u0 < 0.3e9, u1 - u0                       # positions x2, velocities, accel scratch (84 MB; a process's first allocations
                                                          # bring runtime pools with them): no workspace of either kernel yet
    assert ctx.kernel_name() == "nbody_force_sym_f32<false>"   # asks for the slots, like the first step
    u2 = used()
    assert 1.2e9 < u2 - u1 < 2.2e9, u2 - u1
    a_sym = ctx.accel(1)
    ctx.close()
    with c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=1e-2, ordered_pairs=True) as k1:
        k1.set_state(q, v, m)
        assert k1.kernel_name().startswith("nbody_force_f32<")
        a_k1 = k1.accel(1)
        assert used() - u0 < 0.6e9
    rows = np.array([0, 4097, n // 2 + 5, n - 1])
    pos, _ = syn.body4_f32(n)
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    for a in (a_sym, a_k1):
        assert (np.abs(a[:, rows] - ref).max(axis=0) / s).max() < TOL_F32
    assert not np.array_equal(a_sym, a_k1)  # two kernels, two summation orders
    # a limit of the caller's own (NB_CFG_WORKSPACE_GIB): K1s still, in batches of superblocks within 1 GiB
    with c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=1e-2, workspace_gib=1) as lim:
        lim.set_state(q, v, m)
        assert lim.kernel_name() == "nbody_force_sym_f32<false>"
        assert used() - u0 < 0.4e9 + (1 << 30)
        a_lim = lim.accel(1)
    assert (np.abs(a_lim[:, rows] - ref).max(axis=0) / s).max() < TOL_F32
    assert np.abs(a_lim - a_sym).max() < 2e-6 * np.abs(a_sym).max() and not np.array_equal(a_lim, a_sym)  # same pairs, other cuts of the sums
    c.selftest_pair_schedule_within(n, 1 << 30)
    # a GPU with 1.2 GB to spare: the library does the same by itself (3/4 of what is free) and says so
    ctx = c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=1e-2)
    ctx.set_state(q, v, m)
    torch.cuda.empty_cache()
    hog = torch.empty(max(0, torch.cuda.mem_get_info(0)[0] - int(1.2 * (1 << 30))), dtype=torch.uint8, device="cuda:0")
    hog2 = None
    try:
        assert ctx.kernel_name() == "nbody_force_sym_f32<false>"
        assert "note:" in ctx.last_error() and "batches of superblocks" in ctx.last_error()
        a_b = ctx.accel(1)
        ctx.close()
        # ... and with 0.5 GB to spare not even batches of 16 superblocks fit (0.6 GB): the context steps with K1 — whose own
        # 0.3 GB of source slices are allocated only now — and says so
        ctx = c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=1e-2)
        ctx.set_state(q, v, m)
        hog2 = torch.empty(max(0, torch.cuda.mem_get_info(0)[0] - (1 << 29)), dtype=torch.uint8, device="cuda:0")
        assert ctx.kernel_name().startswith("nbody_force_f32<")
        assert "note:" in ctx.last_error() and "every ordered pair (K1) instead" in ctx.last_error()
        a_fb = ctx.accel(1)
    finally:
        del hog, hog2
        torch.cuda.empty_cache()
        ctx.close()
    assert (np.abs(a_b[:, rows] - ref).max(axis=0) / s).max() < TOL_F32
    assert np.array_equal(a_fb, a_k1)  # the fallback IS the ordered-pair context
