"""Rehearsal of the sharded path with the REAL kernel: 2 and 4 ranks share the one GPU of the box and exchange through
gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Checks tgt_off != 0 launches,
the ping-pong and the in-place exchange against the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
N, STEPS = 16384, 4


def _run(rank, world, port, path, acc64=False, backend="gloo", overlap=False, exchange="in_place", N=N, STEPS=STEPS,
         shared_pairs=None):
    sys.path.insert(0, ROOT)
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    from nbody_amd.distributed import ShardedSystem, shard_range
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":  # world 1 only: RCCL refuses two ranks on one device
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                device_id=dev)
    elif world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(N, rank, world)
    pos, vel = synthetic.body4_f32(N, lo, hi)
    if acc64:  # fp64 masters start from the fp64 bodies (what bench.py --precision f32acc64 does)
        q, v, m = synthetic.bodies(N, lo, hi)
        pos = np.concatenate([q.T, (synthetic.G * m)[:, None]], axis=1)
        vel = np.concatenate([v.T, np.zeros((hi - lo, 1))], axis=1)
    sysm = ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, dev, acc64=acc64,
                         overlap=overlap, exchange=exchange, shared_pairs=shared_pairs)
    if shared_pairs is not None:
        assert sysm.shared_pairs == shared_pairs
    if backend == "nccl":
        assert sysm.exchange_mode == "in_place"
        before = sysm.positions.clone()
        dist.all_gather_into_tensor(sysm.positions, sysm.positions[sysm.lo:sysm.hi])  # the aliased call by itself
        torch.cuda.synchronize()
        assert torch.equal(before, sysm.positions)
        # the other collective of a step whose ranks share the unordered pairs: reduce_scatter_tensor of float4[N] partial
        # forces (double4 with fp64 accumulation) — with one rank the sum is the input itself
        for fdt in (torch.float32, torch.float64):
            f = torch.arange(N * 4, dtype=fdt, device=dev).reshape(N, 4)
            o = torch.empty_like(f)
            dist.reduce_scatter_tensor(o, f, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            assert torch.equal(o, f)
    def all_velocities():
        mine = (sysm.vel64 if acc64 else sysm.vel).clone()
        if world == 1:
            return mine
        vels = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(vels, mine)
        return torch.cat(vels)

    vel_first = None
    for k in range(STEPS):
        sysm.step()
        if k == 0 and shared_pairs:  # the velocities after the FIRST step too (checked against oracle rows)
            torch.cuda.synchronize()
            vel_first = all_velocities()
    torch.cuda.synchronize()
    if acc64:  # the fp64 masters are the state of this mode: gather them instead of the fp32 copies
        p64 = [torch.zeros_like(sysm.pos64) for _ in range(world)] if world > 1 else [sysm.pos64]
        if world > 1:
            dist.all_gather(p64, sysm.pos64)
        pos64 = torch.cat(p64).cpu().numpy()
    myv = sysm.vel64 if acc64 else sysm.vel
    if world > 1:
        vels = [torch.zeros_like(myv) for _ in range(world)]
        dist.all_gather(vels, myv)
        allv = torch.cat(vels)
    else:
        allv = myv
    full = sysm.positions  # every rank: with exchange="ring" this is a collective (no rank holds all positions)
    if rank == 0:
        extra = dict(pos64=pos64) if acc64 else {}
        if vel_first is not None:
            extra["vel_first"] = vel_first.cpu().numpy()
        np.savez(path, pos=full.cpu().numpy(), vel=allv.cpu().numpy(), **extra)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _same_to_fp32_rounding(a, b):
    """Two fp32 trajectories of the same system whose launches sum in different orders: all but a handful of bodies within an
    ulp or two, the handful (closest neighbour rounded the other way) within the amplification bound."""
    dq, dv = np.abs(a["pos"] - b["pos"]).max(axis=1), np.abs(a["vel"] - b["vel"]).max(axis=1)
    assert np.quantile(dq, 0.999) < 2.5e-7 and np.quantile(dv, 0.999) < 2e-6, (np.quantile(dq, 0.999), np.quantile(dv, 0.999))
    assert dq.max() < 5e-6 and dv.max() < 2e-4, (dq.max(), dv.max())


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_match_single(nb, tmp_path, world):
    port = _free_port()
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(world, port, two), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    # the sharded launch may pick another register blocking / source split: equal to fp32 rounding, not bitwise.  Nearly every body
    # agrees to an ulp; a body whose closest neighbour's fp32 position rounded the other way in an earlier step sees a force that
    # differs by up to G*m/eps^3 * 6e-8 ~ 4e-3, i.e. 4e-5 per step in v at dt = 1e-2 (the same trajectory-level effect the
    # fp64-accumulate test below spells out) — bounded, and rare
    _same_to_fp32_rounding(a, b)
    p0, _ = nb.synthetic.body4_f32(N)
    assert np.abs(b["pos"][:, :3] - p0[:, :3]).max() > 1e-6 and np.array_equal(b["pos"][:, 3], p0[:, 3])


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_match_single_acc64(nb, oracle, tmp_path, world):
    """NB_F32_ACC64 sharded (BASELINE configs[4]'s arithmetic; bench.py --precision f32acc64 --gpus P): tgt_off != 0
    launches with fp64 masters must move bodies like the unsharded run, and both like the fp64 oracle."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_run, args=(1, 0, one, True), nprocs=1, join=True)
    mp.spawn(_run, args=(world, _free_port(), two, True), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    # fp64 masters: the pair arithmetic is fp32 in both, only the register blocking / slice order may differ
    assert np.abs(a["pos64"] - b["pos64"])[:, :3].max() < 1e-9 and np.abs(a["vel"] - b["vel"]).max() < 1e-7
    assert np.array_equal(a["pos64"][:, 3], b["pos64"][:, 3])
    # the oracle, stepped the way this mode is defined: pair arithmetic on the fp32 copy of the positions, fp64 masters
    # integrated with the fp64 sums (samples/nbody.cc:76-88 for the update)
    syn = nb.synthetic
    q, v, m = syn.bodies(N)
    gm = (syn.G * m).astype(np.float32).astype(np.float64) / syn.G
    for _ in range(STEPS):
        acc = oracle.accel_rows(q.astype(np.float32).astype(np.float64), gm, syn.G, syn.EPS)
        v = v + acc * 1e-2
        q = q + v * 1e-2
    # Trajectory-level bound, not a per-launch one (that is tests/test_gpu_f32_parity.py::test_config4_*): a master that
    # sits within 1e-10 of an fp32 rounding boundary may round the other way on the GPU, which moves that body's copy by
    # 6e-8 and its closest neighbour's force by up to G*m/eps^3 * 6e-8 ~ 1e-3, i.e. 1e-7 per step in q at dt = 1e-2.
    # The bodies move by ~1e-3 over the 4 steps.
    dq = np.abs(b["pos64"][:, :3].T - q).max()
    dv = np.abs(b["vel"][:, :3].T - v).max()
    assert dq < 2e-6 and dv < 2e-4, (dq, dv)
    assert np.abs(b["pos64"][:, :3].T - syn.bodies(N)[0]).max() > 1e-4  # they did move


@pytest.mark.parametrize("world,acc64", [(2, False), (4, True)])
def test_ranks_sharing_the_unordered_pairs_match_single(nb, oracle, tmp_path, world, acc64):
    """shared_pairs (the default from 28672 bodies on, whole 4096-body superblocks per shard): every rank runs K1s on its
    share of the UNORDERED pairs, the partial forces on all N bodies are summed to their owners (reduce-scatter; through
    host memory in this gloo rehearsal), the owner kicks and drifts.  Two steps of N = 2^18 against the one-rank run (K1s on
    the whole system) and the first step's velocities against oracle rows of every shard."""
    n, steps = 1 << 18, 2
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_run, args=(1, 0, one, acc64, "gloo", False, "in_place", n, steps), nprocs=1, join=True)
    mp.spawn(_run, args=(world, _free_port(), two, acc64, "gloo", False, "in_place", n, steps, True), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    if acc64:
        assert np.abs(a["pos64"] - b["pos64"])[:, :3].max() < 1e-9 and np.abs(a["vel"] - b["vel"]).max() < 1e-7
    else:
        _same_to_fp32_rounding(a, b)
    # the first of the two steps against the oracle, rows from every shard
    c = {"vel": b["vel_first"]}
    syn = nb.synthetic
    q, v, m = syn.bodies(n)
    per = n // world
    rows = np.concatenate([[r * per, r * per + per // 2 + 3, (r + 1) * per - 1] for r in range(world)])
    gm = (syn.G * m).astype(np.float32).astype(np.float64) / syn.G
    q32 = q.astype(np.float32).astype(np.float64)
    dt = np.float64(np.float32(1e-2))
    ref_all, s_all = oracle.accel_rows_at(q32, gm, syn.G, syn.EPS, rows, want_abs=True)
    for k, i in enumerate(rows):
        ref, s = ref_all[:, k:k + 1], s_all[k:k + 1]
        v0 = v[:, i] if acc64 else v[:, i].astype(np.float32).astype(np.float64)
        a_gpu = (c["vel"][i, :3].astype(np.float64) - v0) / dt
        slack = 0.0 if acc64 else 2.0 ** -23 * np.abs(c["vel"][i, :3]).max() / dt
        assert np.abs(a_gpu - ref[:, 0]).max() <= (1e-6 if acc64 else 1e-5) * s[0] + slack, (i, a_gpu, ref[:, 0])


def test_nccl_world1_in_place_all_gather(nb, tmp_path):
    """The one RCCL aliasing check reachable on a one-GPU box: a world-size-1 `nccl` group runs the real
    all_gather_into_tensor(buf, buf[lo:hi]) (send buffer = own slot of the receive buffer) every step; the trajectory
    must equal the run without a process group bit for bit (same shard, same plan)."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "nccl.npz")
    mp.spawn(_run, args=(1, 0, one), nprocs=1, join=True)
    mp.spawn(_run, args=(1, _free_port(), two, False, "nccl"), nprocs=1, join=True)
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["vel"], b["vel"])


@pytest.mark.parametrize("world,acc64", [(2, False), (4, False), (4, True)])
def test_overlapped_step_on_one_gpu_matches_single(nb, tmp_path, world, acc64):
    """SURVEY §8(f)-3 with the real kernel: every rank cuts its step into phases (own shard / the shards before it / the
    shards after it: NB_PHASE_FIRST, MIDDLE, LAST with the running sums in the workspace) around the asynchronous
    exchange.  Same trajectory as the unsharded run, to the rounding of a reordered sum."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_run, args=(1, 0, one, acc64), nprocs=1, join=True)
    mp.spawn(_run, args=(world, _free_port(), two, acc64, "gloo", True), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    if acc64:
        assert np.abs(a["pos64"] - b["pos64"])[:, :3].max() < 1e-9 and np.abs(a["vel"] - b["vel"]).max() < 1e-7
    else:
        _same_to_fp32_rounding(a, b)
    p0, _ = nb.synthetic.body4_f32(N)
    assert np.abs(b["pos"][:, :3] - p0[:, :3]).max() > 1e-6 and np.array_equal(b["pos"][:, 3], p0[:, 3])


@pytest.mark.parametrize("world,acc64", [(2, False), (4, False), (4, True)])
def test_ring_pass_on_one_gpu_matches_single(nb, tmp_path, world, acc64):
    """exchange="ring" with the real kernel: the sources of every phase are a travelling block of N/P bodies, the targets
    a separate array (nb_launch_f32.tgt), P phases per step with the running sums in the workspace; no rank holds all N
    positions.  Same trajectory as the unsharded run, to the rounding of a reordered sum."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    mp.spawn(_run, args=(1, 0, one, acc64), nprocs=1, join=True)
    mp.spawn(_run, args=(world, _free_port(), two, acc64, "gloo", False, "ring"), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    if acc64:
        assert np.abs(a["pos64"] - b["pos64"])[:, :3].max() < 1e-9 and np.abs(a["vel"] - b["vel"]).max() < 1e-7
    else:
        _same_to_fp32_rounding(a, b)
    p0, _ = nb.synthetic.body4_f32(N)
    assert np.abs(b["pos"][:, :3] - p0[:, :3]).max() > 1e-6 and np.array_equal(b["pos"][:, 3], p0[:, 3])


@pytest.mark.parametrize("n", [8192 + 256, 131072 + 256])  # K1 (every ordered pair) / K1s (every unordered pair once)
@pytest.mark.parametrize("acc64", [False, True])
def test_sharded_checkpoint_resume_is_bitwise_on_gpu(nb, tmp_path, acc64, n):
    """ShardedSystem.save_checkpoint / load_checkpoint_shard with the real kernels (one rank; the two-rank form runs on CPU
    in tests/test_distributed_gloo.py): 2 steps + checkpoint + a NEW system from the file + 2 steps = 4 uninterrupted steps,
    bit for bit, in NB_F32 and with fp64 masters."""
    from nbody_amd.distributed import ShardedSystem
    syn, c = nb.synthetic, nb.capi
    dev = torch.device("cuda", 0)

    def fresh():
        if acc64:
            q, v, m = syn.bodies(n)
            pos = np.ascontiguousarray(np.concatenate([q.T, (syn.G * m)[:, None]], axis=1))
            vel = np.ascontiguousarray(np.concatenate([v.T, np.zeros((n, 1))], axis=1))
        else:
            pos, vel = syn.body4_f32(n)
        return ShardedSystem(n, torch.from_numpy(pos), torch.from_numpy(vel), syn.EPS, 1e-2, dev, acc64=acc64)

    def state(s):
        torch.cuda.synchronize()
        return (s.pos64 if acc64 else s.positions).cpu().numpy().copy(), (s.vel64 if acc64 else s.vel).cpu().numpy().copy()

    ref = fresh()
    for _ in range(4):
        ref.step()
    a = fresh()
    for _ in range(2):
        a.step()
    path = str(tmp_path / "ck.nbst")
    a.save_checkpoint(path, 2, syn.G)
    assert c.state_file_info(path) == (n, c.NB_F32_ACC64 if acc64 else c.NB_F32, 2)
    hdr, pos, vel = ShardedSystem.load_checkpoint_shard(path, 0, 1)
    assert (hdr["dt"], hdr["eps"]) == (1e-2, syn.EPS)
    if not acc64:
        pos, vel = pos.astype(np.float32), vel.astype(np.float32)
    b = ShardedSystem(n, torch.from_numpy(pos), torch.from_numpy(vel), hdr["eps"], hdr["dt"], dev, acc64=acc64)
    for _ in range(2):
        b.step()
    (p0, v0), (p1, v1) = state(ref), state(b)
    assert np.array_equal(p0[:, :3], p1[:, :3]) and np.array_equal(v0[:, :3], v1[:, :3])
    assert np.array_equal(ref.positions.cpu().numpy(), b.positions.cpu().numpy())  # incl. the G*m column of the fp32 records
