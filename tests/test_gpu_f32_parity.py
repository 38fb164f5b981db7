"""GPU parity of the fp32 / fp32+fp64acc large-N path against the fp64 oracle — through the C ABI.

Tolerance (stated here): the net force on a uniform cloud cancels heavily, so errors are measured
against the sum of magnitudes  S_i = sum_j |a_ij|  the oracle returns:   |a_gpu - a_ref|_inf <= 1e-5 * S_i  (fp32),
<= 1e-6 * S_i (fp32 pair math with fp64 accumulation).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_F32, TOL_ACC64 = 1e-5, 1e-6


def _accel_err(nb, oracle, n, precision, rows=None):
    syn = nb.synthetic
    q, v, m = syn.bodies(n)
    with nb.capi.Context(n, precision, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
        ctx.set_state(q, v, m)
        a = ctx.accel(1)
    q32 = q.astype(np.float32).astype(np.float64)  # the GPU sees fp32-rounded positions: compare like with like
    gm = (syn.G * m).astype(np.float32).astype(np.float64) / syn.G
    if rows is None:
        ref, s = oracle.accel_rows(q32, gm, syn.G, syn.EPS, want_abs=True)
        if n == 1:  # a lone body: no pairs, S = 0, the acceleration must be exactly zero
            assert not a.any()
            return 0.0, a, m
        return (np.abs(a - ref).max(axis=0) / s).max(), a, m
    # an int = one target row, a (first, count) pair = a block of consecutive rows; all of them in ONE oracle call (its per-row
    # loop, OpenMP over the list)
    if len(rows) == 0:  # (the caller only wants the accelerations)
        return 0.0, a, m
    idx = np.concatenate([np.arange(int(i), int(i) + 1) if np.ndim(i) == 0 else np.arange(int(i[0]), int(i[0]) + int(i[1])) for i in rows])
    ref, s = oracle.accel_rows_at(q32, gm, syn.G, syn.EPS, idx, want_abs=True)
    return (np.abs(a[:, idx] - ref).max(axis=0) / s).max(), a, m


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1000, 4096, 16384 + 77])
def test_accel_f32_small_and_ragged(nb, oracle, n):
    err, _, _ = _accel_err(nb, oracle, n, nb.capi.NB_F32)
    assert err < TOL_F32, (n, err)


@pytest.mark.parametrize("n,slices", [(1536, 6), (3584, 14), (6144, 24), (10240 + 5, 21), (22528, 22), (24576, 32), (28672 - 300, 16)])  # (from 28672 bodies on a context runs K1s)
def test_small_whole_systems_take_the_slices_the_model_picks(nb, oracle, n, slices):
    """Round 5: a whole system below K1s' threshold cuts its sources by the measured co-residency model (plan_f32,
    profiles/r05_k1_small_n_model.txt), not into a fixed 16 slices — ragged last tiles and slice counts that do not divide the
    tiles included.  Every row against the oracle, in both arithmetic modes; the slice count itself on a 256-CU device."""
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert nb.capi.plan_f32(n, n, workspace_bytes=66 * n * 16)[1] == slices
    err, _, _ = _accel_err(nb, oracle, n, nb.capi.NB_F32)
    assert err < TOL_F32, (n, err)
    err, _, _ = _accel_err(nb, oracle, n, nb.capi.NB_F32_ACC64)
    assert err < TOL_ACC64, (n, err)


@pytest.mark.parametrize("n", [257, 4096, 16384 + 77])
def test_accel_f32_acc64(nb, oracle, n):
    err, _, _ = _accel_err(nb, oracle, n, nb.capi.NB_F32_ACC64)
    assert err < TOL_ACC64, (n, err)


@pytest.mark.parametrize("path", [1, 2], ids=["lds", "sgpr"])
@pytest.mark.parametrize("tpl,js,acc64", [(2, 1, False), (4, 1, False), (8, 1, False), (4, 2, False), (2, 16, False),
                                           (8, 3, False), (4, 4, True), (8, 1, True), (0, 0, False), (0, 0, True),
                                           (4, 33, False), (8, 20, True)])  # > 16 slices: several launches + running sum
def test_raw_launch_all_register_blockings_and_splits(nb, oracle, tpl, js, acc64, path):
    """nb_launch_accel_f32 on torch-owned HBM: every targets-per-lane variant, source splits with the partial-sum
    reducer, a target window inside the sources."""
    import torch
    syn = nb.synthetic
    n, off, cnt = 8192 + 13, 1000, 3333
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    acc = torch.zeros((cnt, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
    ws = torch.empty(nb.capi.workspace_bytes_f32(cnt, acc64), dtype=torch.uint8, device="cuda")
    nb.capi.launch_f32(src.data_ptr(), 0, n, off, cnt, syn.EPS ** 2, syn.DT,
                       torch.cuda.current_stream().cuda_stream, acc_ptr=acc.data_ptr(), targets_per_lane=tpl,
                       j_split=js, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), acc64=acc64,
                       source_path=path, accel_only=True)
    torch.cuda.synchronize()
    if tpl == 0:
        r, j, w = nb.capi.plan_f32(n, cnt, acc64, workspace_bytes=ws.numel())
        assert r in (2, 4, 8) and 1 <= j <= 1024 and w in (256, 512, 1024)
    a = acc.cpu().numpy()[:, :3].T.astype(np.float64)
    q32 = pos[:, :3].T.astype(np.float64).copy()
    gm = pos[:, 3].astype(np.float64) / syn.G
    ref, s = oracle.accel_rows(q32, gm, syn.G, syn.EPS, off, off + cnt, want_abs=True)
    assert (np.abs(a - ref).max(axis=0) / s).max() < (TOL_ACC64 if acc64 else TOL_F32)


@pytest.mark.parametrize("wg,tpl,js", [(1024, 4, 1), (1024, 4, 4), (512, 8, 1), (512, 8, 2)])
def test_large_workgroup_variants(nb, oracle, wg, tpl, js):
    """SGPR path with one big workgroup per CU (1024 x R=4, 512 x R=8), with and without a source split; ragged n."""
    import torch
    syn = nb.synthetic
    n = 3 * 4096 + 1234
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    acc = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    ws = torch.empty(nb.capi.workspace_bytes_f32(n), dtype=torch.uint8, device="cuda")
    nb.capi.launch_f32(src.data_ptr(), 0, n, 0, n, syn.EPS ** 2, syn.DT, torch.cuda.current_stream().cuda_stream,
                       acc_ptr=acc.data_ptr(), targets_per_lane=tpl, j_split=js, wg_size=wg, source_path=2,
                       workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), accel_only=True)
    torch.cuda.synchronize()
    assert nb.capi.plan_f32(n, n, targets_per_lane=tpl, j_split=js, wg_size=wg, workspace_bytes=ws.numel()) == (tpl, js, wg)
    a = acc.cpu().numpy()[:, :3].T.astype(np.float64)
    q32 = pos[:, :3].T.astype(np.float64).copy()
    ref, s = oracle.accel_rows(q32, pos[:, 3].astype(np.float64) / syn.G, syn.G, syn.EPS, want_abs=True)
    assert (np.abs(a - ref).max(axis=0) / s).max() < TOL_F32


def test_split_step_equals_unsplit_step(nb):
    """The j-split path (partials + reducer + kick-drift) must move bodies like the fused single-kernel path."""
    import torch
    syn = nb.synthetic
    n = 4096 + 5
    pos, vel = syn.body4_f32(n)
    outs = []
    for js, path in ((1, 2), (4, 2), (1, 1), (3, 1)):
        src = torch.from_numpy(pos).cuda()
        out = torch.zeros_like(src)
        v = torch.from_numpy(vel).cuda()
        ws = torch.empty(nb.capi.workspace_bytes_f32(n), dtype=torch.uint8, device="cuda")
        nb.capi.launch_f32(src.data_ptr(), out.data_ptr(), n, 0, n, syn.EPS ** 2, 1e-2,
                           torch.cuda.current_stream().cuda_stream, vel_ptr=v.data_ptr(), targets_per_lane=4,
                           j_split=js, source_path=path, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
        torch.cuda.synchronize()
        outs.append((out.cpu().numpy(), v.cpu().numpy()))
    assert np.array_equal(outs[0][0][:, 3], pos[:, 3])
    for o in outs[1:]:
        assert np.abs(outs[0][0] - o[0]).max() < 1e-6 and np.abs(outs[0][1] - o[1]).max() < 1e-5
    assert not np.array_equal(outs[0][0][:, :3], pos[:, :3])


@pytest.mark.parametrize("precision", ["F32", "F32_ACC64"])
def test_steps_follow_oracle(nb, oracle, precision):
    """20 fused force+kick+drift steps vs the fp64 oracle on the same fp32-rounded start."""
    syn = nb.synthetic
    n = 2048
    q, v, m = syn.bodies(n)
    prec = getattr(nb.capi, "NB_" + precision)
    with nb.capi.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=syn.DT) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 20)
        qg, vg = ctx.get_state()
    s = oracle.System(n)
    if precision == "F32":
        s.q[:] = q.astype(np.float32)
        s.v[:] = v.astype(np.float32)
    else:
        s.q[:], s.v[:] = q, v
    s.m[:] = (syn.G * m).astype(np.float32).astype(np.float64) / syn.G
    oracle.run_steps(s, 1, 20, params=oracle.make_params(dt=syn.DT, eps=syn.EPS, G=syn.G), omp=True)
    # F32 keeps q in fp32: every drift rounds to 0.5 ulp(1.0) = 6e-8, so 20 steps may differ by up to 1.2e-6
    tol = 1.5e-6 if precision == "F32" else 2e-9
    assert np.abs(qg - s.q).max() < tol, np.abs(qg - s.q).max()
    assert np.abs(vg - s.v).max() < (1e-6 if precision == "F32" else 1e-7)


def test_full_size_properties_n2e20(nb, oracle):
    """BASELINE configs[2] size (N=2^20): a full CPU re-run is impossible (1.1e12 pairs), so
    (a) K = 4096 targets — 64 strided blocks of 64, incl. the first and the last bodies — are checked against
    the oracle (4.3e9 pairs on the host's cores: a few seconds with the OpenMP build),
    (b) Newton's third law: sum_i m_i a_i ~ 0, (c) two launches give identical bits."""
    n = 1 << 20
    rows = [(b * (n // 64) + (17 if 0 < b < 63 else 0 if b == 0 else n // 64 - 64), 64) for b in range(64)]
    err, a, m = _accel_err(nb, oracle, n, nb.capi.NB_F32, rows=rows)
    assert err < TOL_F32, err
    p = (a * m).sum(axis=1)
    scale = (np.abs(a) * m).sum(axis=1)
    assert np.all(np.abs(p) < 1e-4 * scale), (p, scale)
    _, a2, _ = _accel_err(nb, oracle, n, nb.capi.NB_F32, rows=[])
    assert np.array_equal(a, a2)


def test_spot_check_n2e22(nb, oracle):
    """BASELINE configs[3] size (N=2^22, the 8-GPU case) on one GPU, fp32 pair math with fp64 accumulation
    (configs[4]'s arithmetic): 64-bit indexing, 16 strided targets against all 4.2M sources vs the oracle."""
    n = 1 << 22
    rows = np.arange(16) * (n // 16) + 5
    err, a, m = _accel_err(nb, oracle, n, nb.capi.NB_F32_ACC64, rows=rows)
    assert err < TOL_ACC64, err
    assert np.isfinite(a).all()


def test_fp32_rejects_eps_zero_and_devices(nb):
    with pytest.raises(nb.capi.NBodyError):
        nb.capi.Context(16, nb.capi.NB_F32, 0, eps=0.0)
    with pytest.raises(nb.capi.NBodyError):
        nb.capi.Context(16, nb.capi.NB_F32, 0, eps=1e-23)  # eps^2 underflows fp32: the self pair would be 0 * inf
    with nb.capi.Context(16, nb.capi.NB_F32, 0) as ctx:
        q = np.zeros((3, 16)); q[0] = np.arange(16)
        with pytest.raises(nb.capi.NBodyError):
            ctx.set_state(q, q, np.ones(16), np.ones(16, dtype=np.uint8))


# ---------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] / configs[4]: the PER-RANK launch shapes of the 8-GPU runs, on one GPU, against the oracle.
# A rank of P = 8 holds all n_src sources and owns n_tgt = n_src / 8 targets starting at tgt_off = rank * n_tgt.

def _oracle_rows(oracle, syn, pos, rows):
    """fp64 reference accelerations + sum of magnitudes for single target rows, from the fp32 records the GPU sees (all rows
    in one call: the oracle's per-row loop, OpenMP over the list)."""
    q32 = np.ascontiguousarray(pos[:, :3].T.astype(np.float64))
    gm = pos[:, 3].astype(np.float64) / syn.G
    return oracle.accel_rows_at(q32, gm, syn.G, syn.EPS, [int(i) for i in rows], want_abs=True)


def _check_step_rows(pos, vel0, out, vel1, rows, off, ref, s, dt, tol):
    """rows of one fused force+kick+drift launch against  v' = v + a*dt ; q' = q + v'*dt  (samples/nbody.cc:76-88)."""
    dt32 = np.float64(np.float32(dt))
    for c, i in enumerate(rows):
        a_gpu = (vel1[i - off, :3].astype(np.float64) - vel0[i - off, :3].astype(np.float64)) / dt32
        # v' is stored in fp32: a is recovered to ulp(v')/dt
        slack = 2.0 ** -23 * np.abs(vel1[i - off, :3]).max() / dt32
        assert np.abs(a_gpu - ref[:, c]).max() <= tol * s[c] + slack, (i, a_gpu, ref[:, c])
        q_exp = (vel1[i - off, :3].astype(np.float64) * dt32 + pos[i, :3].astype(np.float64)).astype(np.float32)
        assert np.abs(out[i, :3] - q_exp).max() <= np.spacing(np.abs(q_exp).max()), (i, out[i], q_exp)
        assert out[i, 3] == pos[i, 3]


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_config3_rank_shape_f32(nb, oracle, rank):
    """configs[3]: N = 2^22, fp32, P = 8  ->  n_src = 2^22, n_tgt = 2^19, tgt_off = rank * 2^19.  Accelerations of 8
    strided rows of the shard, and (rank 3) one fused step, with the plan the sharded host gets by default."""
    import torch
    syn = nb.synthetic
    n, per = 1 << 22, 1 << 19
    off = rank * per
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(nb.capi.workspace_bytes_f32(per), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((per, 4), dtype=torch.float32, device="cuda")
    nb.capi.launch_f32(src.data_ptr(), 0, n, off, per, syn.EPS ** 2, syn.DT, stream, acc_ptr=acc.data_ptr(),
                       workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), accel_only=True)
    torch.cuda.synchronize()
    tpl, js, wg = nb.capi.plan_f32(n, per, workspace_bytes=ws.numel())
    assert (tpl, wg) == (8, 512) and js >= 16  # the sliced 512 x R8 plan of the bench, more slices for the shard
    rows = off + np.arange(8) * (per // 8) + 11
    rows[-1] = off + per - 1  # the shard's last body
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    a = acc.cpu().numpy()[rows - off, :3].T.astype(np.float64)
    assert (np.abs(a - ref).max(axis=0) / s).max() < TOL_F32
    if rank == 3:
        dt = 1e-2
        _, vel_np = syn.body4_f32(n, off, off + per)
        vel = torch.from_numpy(vel_np).cuda()
        out = torch.zeros_like(src)
        nb.capi.launch_f32(src.data_ptr(), out.data_ptr(), n, off, per, syn.EPS ** 2, dt, stream,
                           vel_ptr=vel.data_ptr(), workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        _check_step_rows(pos, vel_np, o, vel.cpu().numpy(), rows, off, ref, s, dt, TOL_F32)
        assert not o[:off].any() and not o[off + per:].any()  # only the rank's own slot is written


def test_config4_rank_shape_acc64(nb, oracle):
    """configs[4]: N = 2^24, fp32 pair math / fp64 accumulate, P = 8  ->  n_src = 2^24, n_tgt = 2^21,
    tgt_off = 5 * 2^21: one fused step of the shard (fp64 masters) against oracle accelerations of 8 rows."""
    import torch
    syn = nb.synthetic
    n, per, rank = 1 << 24, 1 << 21, 5
    off = rank * per
    pos, _ = syn.body4_f32(n)
    src = torch.from_numpy(pos).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(nb.capi.workspace_bytes_f32(per, True), dtype=torch.uint8, device="cuda")
    rows = off + np.arange(8) * (per // 8) + 7
    rows[-1] = off + per - 1
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    # (round 5: the accel-only launch of the same shape is gone — 7.6 s of kernel for what the fused step below shows too:
    # its fp64 velocity masters give back the accelerations exactly, v' = v + a*dt at TOL_ACC64)
    # one fused step of the shard: fp64 masters integrate, the fp32 copy goes to the rank's slot of `out`
    dt = 1e-2
    q, v, m = syn.bodies(n, off, off + per)
    # (np.concatenate of transposed views yields a column-major array: the kernel wants C-ordered double4 records)
    p64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([q.T, (syn.G * m)[:, None]], axis=1))).cuda()
    v64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([v.T, np.zeros((per, 1))], axis=1))).cuda()
    assert p64.is_contiguous() and v64.is_contiguous()
    out = torch.zeros_like(src)
    nb.capi.launch_f32(src.data_ptr(), out.data_ptr(), n, off, per, syn.EPS ** 2, dt, stream, acc64=True,
                       pos64_ptr=p64.data_ptr(), vel64_ptr=v64.data_ptr(), workspace_ptr=ws.data_ptr(),
                       workspace_bytes=ws.numel())
    torch.cuda.synchronize()
    r = rows - off
    dt32 = np.float64(np.float32(dt))
    v_new = v64.cpu().numpy()[r, :3]
    p_new = p64.cpu().numpy()[r, :3]
    v_exp = v.T[r] + ref.T * dt32
    assert np.all(np.abs(v_new - v_exp) <= TOL_ACC64 * s[:, None] * dt32 + 1e-15)
    assert np.all(np.abs(p_new - (q.T[r] + v_new * dt32)) <= 4e-16)
    o = out[off:off + per].cpu().numpy()[r]
    assert np.array_equal(o[:, :3], p_new.astype(np.float32)) and np.array_equal(o[:, 3], pos[rows, 3])


def test_bench_plan_step_against_oracle(nb, oracle):
    """K1 in the shape bench.py timed until round 3 — nbody_force_f32<4,false,false,true,true,512> (512-thread workgroups,
    R = 8, source slices, step mode) + its reducer's kick-drift — stepped once against the oracle on a ragged N = 131072 + 77
    (a raw launch with K1's own workspace: a context of this size runs K1s since round 4, tests/test_gpu_f32_symmetric.py),
    40 rows of q,v vs  v + a_oracle*dt , q + v'*dt  (samples/nbody.cc:76-88)."""
    syn = nb.synthetic
    n, dt = 131072 + 77, 1e-2
    ws = nb.capi.workspace_bytes_f32(n)
    assert nb.capi.kernel_name_f32(n, n, workspace_bytes=ws) == "nbody_force_f32<4, false, false, true, true, 512>"
    assert nb.capi.kernel_name_f32(1 << 20, 1 << 20, workspace_bytes=nb.capi.workspace_bytes_f32(1 << 20)) == \
        "nbody_force_f32<4, false, false, true, true, 512>"
    import torch
    pos, vel = syn.body4_f32(n)
    src, v_gpu = torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda()
    out_gpu = torch.zeros_like(src)
    w = torch.empty(ws, dtype=torch.uint8, device="cuda")
    nb.capi.launch_f32(src.data_ptr(), out_gpu.data_ptr(), n, 0, n, syn.EPS ** 2, dt, torch.cuda.current_stream().cuda_stream,
                       vel_ptr=v_gpu.data_ptr(), workspace_ptr=w.data_ptr(), workspace_bytes=w.numel())
    torch.cuda.synchronize()
    rows = np.concatenate([np.arange(36) * (n // 36) + 3, [n - 77, n - 76, n - 2, n - 1]])  # incl. the ragged tail block
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    _check_step_rows(pos, vel, out_gpu.cpu().numpy(), v_gpu.cpu().numpy(), rows, 0, ref, s, dt, TOL_F32)


@pytest.mark.parametrize("acc64", [False, True])
def test_phased_step_equals_whole_step(nb, oracle, acc64):
    """A step cut into three launches over disjoint source ranges (NB_PHASE_FIRST / MIDDLE / LAST, running sums in the
    workspace; how the overlapped multi-GPU step consumes its own shard before the gathered ones): same accelerations as
    the oracle, same update as the one-launch step to the rounding of a reordered sum; misuse is refused."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n, off, cnt, dt = 3 * 4096 + 100, 4096, 4096, 1e-2
    pos, _ = syn.body4_f32(n)
    _, vel_np = syn.body4_f32(n, off, off + cnt)
    src = torch.from_numpy(pos).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(c.workspace_bytes_f32(cnt, acc64), dtype=torch.uint8, device="cuda")
    q, v, m = syn.bodies(n, off, off + cnt)
    outs = []
    for ranges in ([(0, 0, c.NB_PHASE_WHOLE)],
                   [(off, off + cnt, c.NB_PHASE_FIRST), (0, off, c.NB_PHASE_MIDDLE), (off + cnt, n, c.NB_PHASE_LAST)]):
        out = torch.zeros_like(src)
        kw = dict(workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), acc64=acc64)
        if acc64:
            p64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([q.T, (syn.G * m)[:, None]], axis=1))).cuda()
            v64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([v.T, np.zeros((cnt, 1))], axis=1))).cuda()
            assert p64.is_contiguous() and v64.is_contiguous()
            kw.update(pos64_ptr=p64.data_ptr(), vel64_ptr=v64.data_ptr())
        else:
            vel = torch.from_numpy(vel_np).cuda()
            kw.update(vel_ptr=vel.data_ptr())
        for b, e, ph in ranges:
            c.launch_f32(src.data_ptr(), out.data_ptr(), n, off, cnt, syn.EPS ** 2, dt, stream, phase=ph, src_begin=b,
                         src_end=e, **kw)
        torch.cuda.synchronize()
        outs.append((out.cpu().numpy(), (v64 if acc64 else vel).cpu().numpy()))
    (o0, v0), (o1, v1) = outs
    assert np.abs(o0 - o1).max() <= 6e-8 and np.abs(v0 - v1).max() <= (1e-12 if acc64 else 1e-8)  # a few fp32 ulps of |v| ~ 0.03
    assert not o1[:off].any() and not o1[off + cnt:].any()
    rows = off + np.array([0, 1, 777, 2048, cnt - 1])
    ref, s = _oracle_rows(oracle, syn, pos, rows)
    dt32 = np.float64(np.float32(dt))
    a_gpu = (v1[rows - off, :3].astype(np.float64) - (v.T if acc64 else vel_np[:, :3].astype(np.float64))[rows - off]) / dt32
    slack = 0 if acc64 else 2.0 ** -23 * np.abs(v1[rows - off, :3]).max() / dt32
    assert np.all(np.abs(a_gpu.T - ref) <= (TOL_ACC64 if acc64 else TOL_F32) * s + slack)
    # refused: a phase without a workspace, a range that is not whole tiles, a range past the end
    base = dict(vel_ptr=1, acc64=False)
    for bad in (dict(phase=c.NB_PHASE_FIRST), dict(src_begin=100, src_end=512, workspace_ptr=ws.data_ptr(),
                                                  workspace_bytes=ws.numel()),
                dict(src_begin=0, src_end=300, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel()),
                dict(src_begin=256, src_end=n + 1, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel())):
        with pytest.raises(c.NBodyError) as e:
            c.launch_f32(src.data_ptr(), src.data_ptr(), n, off, cnt, syn.EPS ** 2, dt, stream, **base, **bad)
        assert e.value.code == c.NB_ERR_INVALID


@pytest.mark.parametrize("acc64", [False, True])
def test_travelling_source_blocks_against_the_oracle(nb, oracle, acc64):
    """The ring pass's launch mode (nb_launch_f32.tgt != NULL): the sources of a launch are a block that is NOT the
    targets' own array, `tgt` is a separate float4[n_tgt], tgt_off (> n_src) only places the result in `out`, and a step
    is NB_PHASE_FIRST / MIDDLE / LAST over three such blocks with the running sums in the workspace.  Accelerations and
    one fused kick-drift against oracle rows (samples/nbody.cc:56-88): the oracle sees targets + the three blocks as one
    system in which the targets are massless, so exactly the block bodies attract them."""
    import torch
    c, syn = nb.capi, nb.synthetic
    blk, n_tgt, tgt_off, dt = 4096, 1000 + 37, 4096 + 1000, 1e-2
    pos, vel = syn.body4_f32(3 * blk + n_tgt)
    tgt_np, vel_np = pos[3 * blk:].copy(), vel[3 * blk:].copy()
    blocks = [torch.from_numpy(pos[k * blk:(k + 1) * blk].copy()).cuda() for k in range(3)]
    tgt = torch.from_numpy(tgt_np).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(c.workspace_bytes_f32(n_tgt, acc64), dtype=torch.uint8, device="cuda")
    phases = (c.NB_PHASE_FIRST, c.NB_PHASE_MIDDLE, c.NB_PHASE_LAST)
    common = dict(workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), acc64=acc64, tgt_ptr=tgt.data_ptr())
    # (1) accelerations
    acc = torch.zeros((n_tgt, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
    for b, ph in zip(blocks, phases):
        c.launch_f32(b.data_ptr(), 0, blk, tgt_off, n_tgt, syn.EPS ** 2, dt, stream, accel_only=True,
                     acc_ptr=acc.data_ptr(), phase=ph, **common)
    torch.cuda.synchronize()
    # oracle: one system [block0, block1, block2, targets], the targets massless
    q = pos[:, :3].T.astype(np.float64).copy()
    m = pos[:, 3].astype(np.float64) / syn.G
    m[3 * blk:] = 0.0
    ref, s = oracle.accel_rows(q, m, syn.G, syn.EPS, 3 * blk, 3 * blk + n_tgt, want_abs=True)
    a = acc.cpu().numpy()[:, :3].T.astype(np.float64)
    tol = TOL_ACC64 if acc64 else TOL_F32
    assert (np.abs(a - ref).max(axis=0) / s).max() < tol
    # (2) one fused step written at tgt_off of an `out` array that is larger than any source block
    out = torch.zeros((tgt_off + n_tgt + 5, 4), dtype=torch.float32, device="cuda")
    kw = dict(common)
    if acc64:
        p64 = torch.from_numpy(tgt_np.astype(np.float64)).cuda()
        v64 = torch.from_numpy(vel_np.astype(np.float64)).cuda()
        kw.update(pos64_ptr=p64.data_ptr(), vel64_ptr=v64.data_ptr())
    else:
        v32 = torch.from_numpy(vel_np).cuda()
        kw.update(vel_ptr=v32.data_ptr())
    for b, ph in zip(blocks, phases):
        c.launch_f32(b.data_ptr(), out.data_ptr(), blk, tgt_off, n_tgt, syn.EPS ** 2, dt, stream, phase=ph, **kw)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert not o[:tgt_off].any() and not o[tgt_off + n_tgt:].any()         # only the window is written
    assert np.array_equal(o[tgt_off:tgt_off + n_tgt, 3], tgt_np[:, 3])      # G*m travels with the record
    dt32 = np.float64(np.float32(dt))
    v0 = vel_np[:, :3].astype(np.float64)
    v_ref = v0 + ref.T * dt32
    q_ref = tgt_np[:, :3].astype(np.float64) + v_ref * dt32
    v_gpu = (v64 if acc64 else v32).cpu().numpy()[:, :3].astype(np.float64)
    q_gpu = (p64.cpu().numpy()[:, :3] if acc64 else o[tgt_off:tgt_off + n_tgt, :3].astype(np.float64))
    ulp_v = 0 if acc64 else 2.0 ** -23 * np.abs(v_ref).max()
    assert np.all(np.abs(v_gpu - v_ref) <= tol * s[:, None] * dt32 + ulp_v)
    assert np.abs(q_gpu - q_ref).max() <= (1e-9 if acc64 else 1.2e-7)       # fp32 drift rounds at ulp(1)/2 = 6e-8
    if acc64:  # the fp32 copy that travels on is the rounded master
        assert np.array_equal(o[tgt_off:tgt_off + n_tgt, :3], p64.cpu().numpy()[:, :3].astype(np.float32))


@pytest.mark.parametrize("acc64", [False, True])
def test_one_launch_of_many_slices_equals_several_launches(nb, acc64):
    """A workspace with more than the minimum of 16 partial-sum slots (up to 64) lets a step of js > 16 source slices go
    out as ONE force launch + ONE reducer instead of js/16 of each: the reducer folds the same slices in the same order,
    so accelerations and the fused update are bit for bit the same — also across the phases of a cut step."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n, off, cnt, js, dt = 16384 + 77, 2048, 4096, 40, 1e-2
    pos, _ = syn.body4_f32(n)
    _, vel_np = syn.body4_f32(n, off, off + cnt)
    src = torch.from_numpy(pos).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    rec = 32 if acc64 else 16
    outs = []
    for slots in (16, 64, 24):
        ws = torch.empty((slots + 2) * cnt * rec, dtype=torch.uint8, device="cuda")
        assert ws.numel() >= c.workspace_bytes_f32(cnt, acc64)
        acc = torch.zeros((cnt, 4), dtype=torch.float64 if acc64 else torch.float32, device="cuda")
        kw = dict(workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), acc64=acc64, j_split=js)
        c.launch_f32(src.data_ptr(), 0, n, off, cnt, syn.EPS ** 2, dt, stream, accel_only=True, acc_ptr=acc.data_ptr(), **kw)
        out = torch.zeros_like(src)
        if acc64:
            q, v, m = syn.bodies(n, off, off + cnt)
            p64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([q.T, (syn.G * m)[:, None]], axis=1))).cuda()
            v64 = torch.from_numpy(np.ascontiguousarray(np.concatenate([v.T, np.zeros((cnt, 1))], axis=1))).cuda()
            kw.update(pos64_ptr=p64.data_ptr(), vel64_ptr=v64.data_ptr())
        else:
            v32 = torch.from_numpy(vel_np).cuda()
            kw.update(vel_ptr=v32.data_ptr())
        for b, e, ph in ((off, off + cnt, c.NB_PHASE_FIRST), (0, off, c.NB_PHASE_MIDDLE), (off + cnt, n, c.NB_PHASE_LAST)):
            c.launch_f32(src.data_ptr(), out.data_ptr(), n, off, cnt, syn.EPS ** 2, dt, stream, phase=ph, src_begin=b,
                         src_end=e, **kw)
        torch.cuda.synchronize()
        outs.append((acc.cpu().numpy(), out.cpu().numpy(), (v64 if acc64 else v32).cpu().numpy()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
    assert outs[0][0].any() and outs[0][1][off:off + cnt].any()
