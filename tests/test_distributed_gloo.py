"""The N>1 path on CPU: world_size-2 `gloo` run of nbody_amd.distributed.ShardedSystem.

The sharding / ping-pong / in-place all-gather logic is what is under test; the per-rank arithmetic is injected
(`compute=`) and is the ORACLE here — test infrastructure standing in for the HIP launch, which the product uses
by default and which refuses CPU tensors (checked below).  Sharded and unsharded runs must agree bit for bit
because each target row is computed by exactly one rank from the same gathered sources.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N, STEPS = 512, 3


def _oracle_compute(O, G, eps):
    carry = {}

    def compute(src, out, vel, off, n_tgt, eps2, dt, pos64=None, vel64=None, src_range=None, phase=0, tgt=None):
        p = src.numpy().astype(np.float64)
        q = np.ascontiguousarray(p[:, :3].T)
        gm = np.ascontiguousarray(p[:, 3] / G)
        if src_range is not None:  # a phase of a step: only these sources pull (the others are made massless)
            mask = np.zeros_like(gm)
            mask[src_range[0]:src_range[1]] = 1.0
            gm = gm * mask
        if tgt is not None:  # ring pass: `src` is a travelling block, the targets are a separate array
            t = tgt.numpy().astype(np.float64)
            d = p[None, :, :3] - t[:, None, :3]                      # (targets, sources, 3); the self pair is d = 0
            w = (G * gm)[None, :] * (np.square(d).sum(axis=2) + eps * eps) ** -1.5
            a = (w[:, :, None] * d).sum(axis=1).T
            p = t  # the update below reads the targets' own records
        else:
            a = O.accel_rows(q, gm, G, eps, off, off + n_tgt, omp=False)
        if phase in (2, 3):   # NB_PHASE_LAST / NB_PHASE_MIDDLE continue the running sum
            a = carry["a"] + a
        if phase in (1, 3):   # NB_PHASE_FIRST / NB_PHASE_MIDDLE keep it for the next phase
            carry["a"] = a
            return
        v = vel.numpy()
        v[:, :3] = (v[:, :3].astype(np.float64) + a.T * dt).astype(np.float32)
        newp = p[off:off + n_tgt, :3] + v[:, :3].astype(np.float64) * dt
        o = out.numpy()
        o[off:off + n_tgt, :3] = newp.astype(np.float32)
        o[off:off + n_tgt, 3] = p[off:off + n_tgt, 3].astype(np.float32)
    return compute


def _oracle_pair_steps(G, eps, n, world):
    """Stand-ins for the two launches of the shared-pairs step (K1s share + kick-drift), same rule at the size of one body:
    the unordered pair {i, j} belongs to body i when (j - i) mod n is in 1..n/2-1, or is n/2 with i in the first half — the
    cyclic schedule of csrc/nbody_kernels.h (sym_rounds) with superblocks of one body — and to the rank whose shard holds i.
    The owner adds +f to i's row and -f to j's row of its partial force on ALL n bodies."""
    i = np.arange(n)[:, None]
    d = (np.arange(n)[None, :] - i) % n
    owned_by_i = ((d >= 1) & (d < n // 2)) | ((d == n // 2) & (i < n // 2)) if n % 2 == 0 else ((d >= 1) & (d <= n // 2))
    assert np.array_equal(owned_by_i | owned_by_i.T, ~np.eye(n, dtype=bool)) and not (owned_by_i & owned_by_i.T).any()

    def pair_forces(src, lo, n_tgt, eps2, fpart):
        p = src.numpy().astype(np.float64)
        dvec = p[None, :, :3] - p[:, None, :3]                       # [i, j] = q_j - q_i
        w = (np.square(dvec).sum(axis=2) + eps * eps) ** -1.5
        pull = w[:, :, None] * dvec                                  # [i, j] * G m_j = acceleration of i towards j
        gm = p[:, 3]
        mine = np.zeros((n, n), dtype=bool)
        mine[lo:lo + n_tgt] = owned_by_i[lo:lo + n_tgt]              # the pairs this rank evaluates
        a = (np.where(mine[:, :, None], pull, 0.0) * gm[None, :, None]).sum(axis=1)          # +f on the owner side
        a -= (np.where(mine[:, :, None], pull, 0.0) * gm[:, None, None]).sum(axis=0)         # -f on the partner's row
        f = fpart.numpy()
        f[:, :3] = a.astype(f.dtype)
        f[:, 3] = 0

    def kick_drift(src, out, vel, lo, n_tgt, dt, facc, pos64, vel64):
        p = src.numpy().astype(np.float64)
        v = vel.numpy()
        v[:, :3] = (v[:, :3].astype(np.float64) + facc.numpy()[:, :3].astype(np.float64) * dt).astype(np.float32)
        o = out.numpy()
        o[lo:lo + n_tgt, :3] = (p[lo:lo + n_tgt, :3] + v[:, :3].astype(np.float64) * dt).astype(np.float32)
        o[lo:lo + n_tgt, 3] = p[lo:lo + n_tgt, 3].astype(np.float32)
    return pair_forces, kick_drift


def _run_shared(rank, world, port, result_path, acc64):
    """The default multi-rank step (ranks share the unordered pairs): partial force on all bodies -> sum over ranks, each
    keeping its shard -> kick-drift -> all-gather; arithmetic injected as above."""
    sys.path.insert(0, ROOT)
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    from nbody_amd.distributed import ShardedSystem, shard_range
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(N, rank, world)
    pos, vel = synthetic.body4_f32(N, lo, hi)
    sysm = ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, torch.device("cpu"),
                         acc64=acc64, pair_steps=_oracle_pair_steps(synthetic.G, synthetic.EPS, N, world))
    assert sysm.shared_pairs and sysm.exchange_mode == "list"
    for _ in range(STEPS):
        sysm.step()
    assert sysm._fpart.dtype == (torch.float64 if acc64 else torch.float32) and sysm._facc.shape == (hi - lo, 4)
    full = sysm.positions.clone()
    ref = full.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, full), "ranks disagree on gathered positions"
    vels = [torch.zeros_like(sysm.vel) for _ in range(world)]
    dist.all_gather(vels, sysm.vel)
    if rank == 0:
        np.savez(result_path, pos=full.numpy(), vel=torch.cat(vels).numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(rank, world, port, result_path, overlap=False, exchange="in_place", ckpt=None):
    sys.path.insert(0, ROOT)
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    from nbody_amd.distributed import ShardedSystem, shard_range
    from oracle import oracle as O
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(N, rank, world)
    pos, vel = synthetic.body4_f32(N, lo, hi)
    sysm = ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, torch.device("cpu"),
                         compute=_oracle_compute(O, synthetic.G, synthetic.EPS), overlap=overlap, exchange=exchange)
    assert (sysm.lo, sysm.hi) == (lo, hi) and sysm.overlap == (overlap and world > 1)
    assert sysm.exchange_mode == ("none" if world == 1 else "ring" if exchange == "ring" else "list")
    if ckpt:  # stop after the first step, write a checkpoint, and continue in a NEW system built from the file
        sysm.step()
        sysm.save_checkpoint(ckpt, 1, synthetic.G)
        hdr, p2, v2 = ShardedSystem.load_checkpoint_shard(ckpt, rank, world)
        assert (hdr["n"], hdr["step"], hdr["dt"], hdr["eps"], hdr["G"]) == (N, 1, 1e-2, synthetic.EPS, synthetic.G)
        sysm = ShardedSystem(N, torch.from_numpy(p2), torch.from_numpy(v2), hdr["eps"], hdr["dt"], torch.device("cpu"),
                             compute=_oracle_compute(O, synthetic.G, synthetic.EPS), overlap=overlap, exchange=exchange)
    for _ in range(STEPS - (1 if ckpt else 0)):
        sysm.step()
    # every rank must hold the same full position array after the exchange
    full = sysm.positions.clone()
    if world > 1:
        ref = full.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, full), "ranks disagree on gathered positions"
        vels = [torch.zeros_like(sysm.vel) for _ in range(world)]
        dist.all_gather(vels, sysm.vel)
        allv = torch.cat(vels)
    else:
        allv = sysm.vel
    if rank == 0:
        np.savez(result_path, pos=full.numpy(), vel=allv.numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_equal_one_rank(oracle, tmp_path):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _run(0, 1, 0, one)
    mp.spawn(_run, args=(2, _free_port(), two), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["vel"], b["vel"])
    # and the sharded run really moved the bodies
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    p0, _ = synthetic.body4_f32(N)
    assert not np.array_equal(p0[:, :3], b["pos"][:, :3]) and np.array_equal(p0[:, 3], b["pos"][:, 3])


@pytest.mark.parametrize("world,acc64", [(2, False), (2, True), (4, False), (8, False)])  # (8: the driver's full node)
def test_ranks_sharing_the_unordered_pairs_equal_one_rank(oracle, tmp_path, world, acc64):
    """What bench.py --gpus P runs by default from two ranks up: every unordered pair evaluated once, by one rank; the
    partial forces summed across the ranks.  Same trajectory as the one-rank run to the rounding of the partial forces
    (fp32 partials: a few ulp of the acceleration; fp64 partials: the last bit of the fp32 state)."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "shared.npz")
    _run(0, 1, 0, one)
    mp.spawn(_run_shared, args=(world, _free_port(), two, acc64), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    scale = np.abs(a["vel"][:, :3]).max()
    tol_v = (1e-9 if acc64 else 3e-6) * max(scale, 1.0)
    assert np.abs(a["vel"] - b["vel"]).max() <= tol_v, np.abs(a["vel"] - b["vel"]).max()
    assert np.abs(a["pos"] - b["pos"]).max() <= (1.2e-7 if acc64 else 5e-7), np.abs(a["pos"] - b["pos"]).max()
    assert np.array_equal(a["pos"][:, 3], b["pos"][:, 3])


def test_shared_pairs_is_refused_where_it_cannot_run():
    """One rank, or the two-phase / ring steps: asking for shared pairs explicitly is an error, not a silent other path."""
    sys.path.insert(0, ROOT)
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    from nbody_amd.distributed import ShardedSystem
    pos, vel = synthetic.body4_f32(N)
    with pytest.raises(ValueError, match="shared_pairs needs"):
        ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, torch.device("cpu"),
                      shared_pairs=True, pair_steps=(None, None))


def test_overlapped_two_phase_step_equals_plain_step(oracle, tmp_path):
    """own-shard sources first, the asynchronous all-gather of the other shards in flight, remote sources
    after it.  Same trajectory as the plain step (the phases only reorder a sum that the stand-in computes in fp64)."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _run(0, 1, 0, one)
    mp.spawn(_run, args=(2, _free_port(), two, True), nprocs=2, join=True)
    a, b = np.load(one), np.load(two)
    assert np.abs(a["pos"] - b["pos"]).max() <= 1.2e-7 and np.abs(a["vel"] - b["vel"]).max() <= 1e-9
    assert np.array_equal(a["pos"][:, 3], b["pos"][:, 3])


@pytest.mark.parametrize("world", [2, 4])
def test_ring_pass_equals_all_gather(oracle, tmp_path, world):
    """exchange="ring": no rank holds all positions; blocks of N/P sources travel rank to rank while the previous one is
    consumed (P phases per step, running sums carried between them).  Same trajectory as the one-rank run."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "ring.npz")
    _run(0, 1, 0, one)
    mp.spawn(_run, args=(world, _free_port(), two, False, "ring"), nprocs=world, join=True)
    a, b = np.load(one), np.load(two)
    assert np.abs(a["pos"] - b["pos"]).max() <= 1.2e-7 and np.abs(a["vel"] - b["vel"]).max() <= 1e-9
    assert np.array_equal(a["pos"][:, 3], b["pos"][:, 3])


@pytest.mark.parametrize("exchange", ["in_place", "ring"])
def test_sharded_checkpoint_resume_is_bitwise(oracle, tmp_path, exchange):
    """SURVEY §8(f)-4 for the sharded host: rank 0 writes one NBODYST2 file for the whole system (velocities gathered from
    their owners), every rank restarts from its slice of it; the resumed trajectory equals the uninterrupted one."""
    ref, res = str(tmp_path / "ref.npz"), str(tmp_path / "res.npz")
    mp.spawn(_run, args=(2, _free_port(), ref, False, exchange), nprocs=2, join=True)
    mp.spawn(_run, args=(2, _free_port(), res, False, exchange, str(tmp_path / "ck.nbst")), nprocs=2, join=True)
    a, b = np.load(ref), np.load(res)
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["vel"], b["vel"])
    import nbody_amd  # noqa: F401
    from nbody_amd import capi
    assert capi.state_file_info(str(tmp_path / "ck.nbst")) == (N, capi.NB_F32, 1)


def test_overlap_needs_tile_aligned_shards(nb):
    from nbody_amd.distributed import ShardedSystem

    class TwoRanks(ShardedSystem):  # shard bookkeeping only: no process group is started
        pass
    import unittest.mock as um
    with um.patch("torch.distributed.is_initialized", return_value=True), \
            um.patch("torch.distributed.get_world_size", return_value=2), \
            um.patch("torch.distributed.get_rank", return_value=0):
        with pytest.raises(ValueError, match="multiple of 256"):
            TwoRanks(600, torch.zeros((300, 4)), torch.zeros((300, 4)), 1e-3, 1e-2, torch.device("cpu"),
                     compute=lambda *a, **k: None, overlap=True)


def test_shard_range_and_cpu_refusal(nb):
    from nbody_amd.distributed import hip_compute, shard_range
    assert [shard_range(1 << 20, r, 8) for r in (0, 7)] == [(0, 131072), (917504, 1048576)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 3)
    t = torch.zeros((8, 4))
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        hip_compute()(t, t.clone(), t.clone(), 0, 8, 1e-6, 1e-4)


def _run_disagreeing(rank, world, port, result_path):
    """Rank 1 cannot share the pairs (no pair launches: as a rank whose device-local workspace query answered 0 would be), the
    others can.  Every rank must end up with the ordered form — never a subset enqueueing the second collective."""
    sys.path.insert(0, ROOT)
    import nbody_amd  # noqa: F401
    from nbody_amd import synthetic
    from nbody_amd.distributed import ShardedSystem, shard_range
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(N, rank, world)
    pos, vel = synthetic.body4_f32(N, lo, hi)
    steps = None if rank == 1 else _oracle_pair_steps(synthetic.G, synthetic.EPS, N, world)
    sysm = ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, torch.device("cpu"),
                         compute=_oracle_compute(O, synthetic.G, synthetic.EPS), pair_steps=steps)
    assert sysm.shared_pairs is False, "a rank-local refusal must switch EVERY rank to the ordered form"
    for _ in range(STEPS):
        sysm.step()
    full = sysm.positions.clone()
    vels = [torch.zeros_like(sysm.vel) for _ in range(world)]
    dist.all_gather(vels, sysm.vel)
    if rank == 0:
        np.savez(result_path, pos=full.numpy(), vel=torch.cat(vels).numpy())
    # asking for shared pairs explicitly where one rank cannot is an error ON EVERY RANK (nobody is left waiting in a collective)
    try:
        ShardedSystem(N, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, 1e-2, torch.device("cpu"),
                      compute=_oracle_compute(O, synthetic.G, synthetic.EPS), pair_steps=steps, shared_pairs=True)
        raised = False
    except ValueError:
        raised = True
    assert raised
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_form_of_the_step(oracle, tmp_path):
    """whether the step shares the pairs depends on device-local facts; the ranks settle it with an all_reduce(MIN).
    Three ranks, one of which cannot: all run the ordered form and reproduce the one-rank trajectory bit for bit."""
    one, three = str(tmp_path / "one.npz"), str(tmp_path / "three.npz")
    _run(0, 1, 0, one)
    # N = 512 bodies over 4 ranks (3 would not divide it)
    mp.spawn(_run_disagreeing, args=(4, _free_port(), three), nprocs=4, join=True)
    a, b = np.load(one), np.load(three)
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["vel"], b["vel"])
